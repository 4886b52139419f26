"""Import the unmodified reference (``/root/reference``) through the stand-in
packages of ``oracle/refstubs``.  Works only in the build container; used by
``oracle/gen_golden.py`` and by the (auto-skipped elsewhere) cross-check tests.

TEST INFRASTRUCTURE — never imported by the product.
"""
import glob
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFROOT = os.environ.get("DRIFT_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFROOT, "drift"))


def load_fast_tools():
    """Load the reference's compiled Cython extension from oracle/_ref (built by
    ``make -C oracle ref``) under its canonical module name."""
    stubs = os.path.join(HERE, "refstubs")
    if stubs not in sys.path:
        sys.path.insert(0, stubs)
    cands = glob.glob(os.path.join(HERE, "_ref", "_fast_tools*.so"))
    if not cands:
        return None
    name = "drift.util._fast_tools"
    if name in sys.modules and getattr(sys.modules[name], "__file__", None):
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, cands[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load():
    """Return a dict of reference modules (beamtransfer, kltransform, doublekl,
    telescope, cylinder, cylbeam, visibility, fast_tools)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFROOT)
    stubs = os.path.join(HERE, "refstubs")
    for p in (REFROOT, stubs):
        if p not in sys.path:
            sys.path.insert(0, p)
    ft = load_fast_tools()
    if ft is None:
        raise RuntimeError("oracle/_ref not built: run `make -C oracle ref`")
    import drift.util  # noqa: F401  (package first, then graft the extension in)

    sys.modules["drift.util._fast_tools"] = ft
    sys.modules["drift.util"]._fast_tools = ft
    from drift.core import beamtransfer, doublekl, kltransform, psestimation, telescope, visibility
    from drift.telescope import cylbeam, cylinder

    return dict(
        beamtransfer=beamtransfer,
        kltransform=kltransform,
        doublekl=doublekl,
        psestimation=psestimation,
        telescope=telescope,
        cylinder=cylinder,
        cylbeam=cylbeam,
        visibility=visibility,
        fast_tools=ft,
    )
