"""Import the *real* reference (read-only at /root/reference) in the build
container -- TEST INFRASTRUCTURE ONLY, never used on the GPU box.

The reference's single native module (tt_sketch/drm/fast_lazy_gaussian.pyx)
is compiled by ``oracle/build_ref.sh`` from the source where it lies into
``oracle/_ref/`` (git-ignored); this loader registers that extension under
its package name and puts /root/reference on ``sys.path``.  Used only by
``tests/golden/make_golden.py`` (fixture generation) and by the optional
"live reference" tests that skip when /root/reference is absent.
"""
import glob
import importlib.machinery
import importlib.util
import os
import sys

REFERENCE_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "tt_sketch")) and bool(
        glob.glob(os.path.join(_HERE, "_ref", "fast_lazy_gaussian*.so")))


def load():
    """Returns the imported ``tt_sketch`` reference package."""
    if not available():
        raise RuntimeError("reference or oracle/_ref build not available")
    sys.dont_write_bytecode = True  # /root/reference is read-only
    name = "tt_sketch.drm.fast_lazy_gaussian"
    if name not in sys.modules:
        so = glob.glob(os.path.join(_HERE, "_ref", "fast_lazy_gaussian*.so"))[0]
        loader = importlib.machinery.ExtensionFileLoader(name, so)
        spec = importlib.util.spec_from_file_location(name, so, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        sys.modules[name] = mod
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import tt_sketch  # noqa: F401
    import tt_sketch.sketch  # noqa: F401
    return sys.modules["tt_sketch"]
