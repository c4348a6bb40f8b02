import ast
import glob
import importlib
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
scenes = importlib.import_module("eradiate-kernel_amd.scenes")


def golden_cases():
    out = []
    for f in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        name = os.path.splitext(os.path.basename(f))[0]
        if not name.startswith("indep_pin_"):                 # fixtures of tests/test_independent_pin.py, another format
            out.append(name)
    return out


def load_golden(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    kwargs = ast.literal_eval(str(z["kwargs"]))
    d = getattr(scenes, str(z["builder"]))(**kwargs)
    return d, z["film"], z["counters"]
