"""Loader for the reference's numpy-only modules (generation time only).

Used ONLY by tests/golden/gen_golden.py, in the build container where
/root/reference is mounted.  `import sitrack` fails there (its __init__ pulls
netCDF4), so util/locate/tracking are mounted under a synthetic package object
so that their lazy relative imports (`from .util import ...`) resolve.
Nothing from the reference is copied: only numeric outputs are written.
"""
import importlib.util
import sys
import types

REF = "/root/reference"


def load_reference(ref=REF):
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("sitrack")
    pkg.__path__ = [ref + "/sitrack"]
    sys.modules["sitrack"] = pkg
    mods = {}
    for name in ("util", "locate", "tracking"):
        spec = importlib.util.spec_from_file_location(
            "sitrack." + name, "%s/sitrack/%s.py" % (ref, name))
        m = importlib.util.module_from_spec(spec)
        sys.modules["sitrack." + name] = m
        spec.loader.exec_module(m)
        setattr(pkg, name, m)
        mods[name] = m
    return mods["util"], mods["locate"], mods["tracking"]
