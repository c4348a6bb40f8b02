"""Mitsuba binary volume files, version 3 (/root/reference/src/textures/volume_data.h:42-102):
"VOL", u8 version = 3, i32 type = 1 (float32), i32 nx, ny, nz, i32 channels, 6 x f32 bbox,
then nz*ny*nx*channels float32 values with x varying fastest (grid3d.cpp:30-34)."""
import struct
import numpy as np


def read_volume(filename):
    with open(filename, "rb") as f:
        header = f.read(3)
        if header != b"VOL":
            raise RuntimeError("Invalid volume file %s" % filename)
        version = struct.unpack("<B", f.read(1))[0]
        if version != 3:
            raise RuntimeError("Invalid version, currently only version 3 is supported (found %d)" % version)
        data_type = struct.unpack("<i", f.read(4))[0]
        if data_type != 1:
            raise RuntimeError("Wrong type, currently only type == 1 (Float32) data is supported (found type = %d)" % data_type)
        nx, ny, nz = struct.unpack("<3i", f.read(12))
        if nx * ny * nz < 8:
            raise RuntimeError("Invalid grid dimensions: %d x %d x %d < 8 (must have at least one value at each corner)" % (nx, ny, nz))
        channels = struct.unpack("<i", f.read(4))[0]
        dims = struct.unpack("<6f", f.read(24))
        count = nx * ny * nz * channels
        data = np.fromfile(f, dtype="<f4", count=count)
        if data.size != count:
            raise RuntimeError("Volume file %s is truncated" % filename)
    meta = {"bbox_min": dims[:3], "bbox_max": dims[3:], "shape": (nx, ny, nz), "channels": channels}
    return data.reshape(nz, ny, nx, channels).astype(np.float32), meta


def write_volume(filename, data, bbox_min=(0.0, 0.0, 0.0), bbox_max=(1.0, 1.0, 1.0)):
    data = np.asarray(data, dtype=np.float32)
    if data.ndim == 3:
        data = data[..., None]
    nz, ny, nx, ch = data.shape
    with open(filename, "wb") as f:
        f.write(b"VOL")
        f.write(struct.pack("<B", 3))
        f.write(struct.pack("<i", 1))
        f.write(struct.pack("<3i", nx, ny, nz))
        f.write(struct.pack("<i", ch))
        f.write(struct.pack("<6f", *(tuple(bbox_min) + tuple(bbox_max))))
        data.astype("<f4").tofile(f)
