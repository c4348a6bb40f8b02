"""CIE standard illuminant D65, relative spectral power every 5 nm over 360 .. 830 nm (95 samples): the published standard data the
`d65` spectrum plugin tabulates (/root/reference/src/spectra/d65.cpp:8-21); the plugin expands to a `regular` spectrum of
these values times scale / 10568 (d65.cpp:52-71)."""

D65 = [
    46.6383, 49.3637, 52.0891, 51.0323, 49.9755, 52.3118, 54.6482, 68.7015, 82.7549, 87.1204,
    91.486, 92.4589, 93.4318, 90.057, 86.6823, 95.7736, 104.865, 110.936, 117.008, 117.41,
    117.812, 116.336, 114.861, 115.392, 115.923, 112.367, 108.811, 109.082, 109.354, 108.578,
    107.802, 106.296, 104.79, 106.239, 107.689, 106.047, 104.405, 104.225, 104.046, 102.023,
    100, 98.1671, 96.3342, 96.0611, 95.788, 92.2368, 88.6856, 89.3459, 90.0062, 89.8026,
    89.5991, 88.6489, 87.6987, 85.4936, 83.2886, 83.4939, 83.6992, 81.863, 80.0268, 80.1207,
    80.2146, 81.2462, 82.2778, 80.281, 78.2842, 74.0027, 69.7213, 70.6652, 71.6091, 72.979,
    74.349, 67.9765, 61.604, 65.7448, 69.8856, 72.4863, 75.087, 69.3398, 63.5927, 55.0054,
    46.4182, 56.6118, 66.8054, 65.0941, 63.3828, 63.8434, 64.304, 61.8779, 59.4519, 55.7054,
    51.959, 54.6998, 57.4406, 58.8765, 60.3125,
]
D65_NORMALIZATION = 1.0 / 10568.0


def _cie_1931_from_header():
    """The CIE 1931 2-degree observer, 95 samples per curve every 5 nm over 360 .. 830 nm: read from the one place this repository keeps
    the table (csrc/cie_tables.h, the array the kernels use; /root/reference/include/mitsuba/core/spectrum.h:127-133 declares it)."""
    import os
    import re
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "cie_tables.h")).read()
    vals = [float(x) for x in re.findall(r"([0-9.eE+-]+)f", text.split("{")[1])]
    assert len(vals) == 3 * 95, len(vals)
    return [vals[0:95], vals[95:190], vals[190:285]]


CIE_1931 = _cie_1931_from_header()
