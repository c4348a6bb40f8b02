"""`import mitsuba_amd as mitsuba` -- root-level alias of the package in ./eradiate-kernel_amd
(whose directory name is not a Python identifier)."""
import importlib as _importlib
import sys as _sys

_pkg = _importlib.import_module("eradiate-kernel_amd")
_sys.modules[__name__] = _pkg
