import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# Box modes of the suite.  The level objects merge the boxes a rank owns (mf.h: coalesce_layout, IAMRX_COALESCE = 1,
# the library's default and what bench.py times).  Every -m gpu test runs in THAT mode first.  A test whose level objects actually merged
# something (iamrx_coalesce_merge_count moved: its result depends on the mode) is then run a second time with the callers' boxes kept
# (IAMRX_COALESCE = 0), so that ghost exchanges between boxes, partial tiles and the per-box multigrid paths stay covered as they were
# when the whole suite ran that way.  Markers: `boxes_kept` -- only the kept mode (tests of the multi-box kernels that say so);
# `merged_only` -- only the default mode; `both_box_modes` -- both, whatever the counter says (tests whose library calls happen in child
# processes: they inherit the mode through IAMRX_COALESCE).  IAMRX_TEST_COALESCE = 0 / 1 pins the whole suite to one mode.
_PIN = os.environ.get("IAMRX_TEST_COALESCE")
if _PIN is not None:
    os.environ["IAMRX_COALESCE"] = _PIN


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run through the HIP C-ABI)")
    config.addinivalue_line("markers", "boxes_kept: run with the callers' boxes kept as boxes only (IAMRX_COALESCE = 0)")
    config.addinivalue_line("markers", "merged_only: run in the library's default box mode only (IAMRX_COALESCE = 1)")
    config.addinivalue_line("markers", "both_box_modes: run in both box modes regardless of the merge counter")


def pytest_collection_modifyitems(config, items):
    # a hung rendezvous or device wait must cost one test, not the driver's whole GPU-test budget (pytest-timeout is in the image)
    if config.pluginmanager.hasplugin("timeout"):
        for it in items:
            if "gpu" in it.keywords and it.get_closest_marker("timeout") is None:
                it.add_marker(pytest.mark.timeout(420))


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def gpu():
    """initialised HIP library; fails loudly (no CPU fallback) when the device or the .so is missing"""
    from iamr_amd import lib
    lib.init(0)
    lib.tuning_set("COALESCE", float(_PIN) if _PIN is not None else 1.0)
    lib.tuning_set("CHECK_UNIFORM", 1)        # arrays marked uniform (MultiFab::mark_uniform) are verified wherever a solver takes the mark
    return lib


def _set_mode(kept):
    from iamr_amd import lib
    lib.tuning_set("COALESCE", 0.0 if kept else 1.0)
    os.environ["IAMRX_COALESCE"] = "0" if kept else "1"


def _merges():
    import ctypes as C
    from iamr_amd import lib
    n = C.c_size_t()
    lib.check(lib.lib().iamrx_coalesce_merge_count(C.byref(n)))
    return n.value


BOX_MODE_LOG = []       # (nodeid, modes run): written to gpurun_out/box_modes.txt at the end of a GPU session


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    if "gpu" not in pyfuncitem.keywords:
        return None
    from iamr_amd import lib
    if getattr(lib, "_lib", None) is None:
        lib.init(0)
    fn = pyfuncitem.obj
    args = {a: pyfuncitem.funcargs[a] for a in pyfuncitem._fixtureinfo.argnames}
    kept_only = pyfuncitem.get_closest_marker("boxes_kept") is not None
    merged_only = pyfuncitem.get_closest_marker("merged_only") is not None
    both = pyfuncitem.get_closest_marker("both_box_modes") is not None
    if _PIN is not None:
        fn(**args)
        return True
    modes = []
    try:
        if kept_only:
            _set_mode(True); modes.append("kept")
            fn(**args)
        else:
            _set_mode(False); modes.append("merged")
            m0 = _merges()
            fn(**args)
            if not merged_only and (both or _merges() > m0):
                _set_mode(True); modes.append("kept")
                try:
                    fn(**args)
                except Exception as e:
                    raise AssertionError(f"[box mode: boxes kept, IAMRX_COALESCE=0] {type(e).__name__}: {e}") from e
    finally:
        _set_mode(False)
        BOX_MODE_LOG.append((pyfuncitem.nodeid, "+".join(modes)))
    return True


def pytest_sessionfinish(session, exitstatus):
    if BOX_MODE_LOG:
        try:
            d = os.path.join(ROOT, "gpurun_out"); os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "box_modes.txt"), "w") as f:
                for nid, m in BOX_MODE_LOG:
                    f.write(f"{m:14s} {nid}\n")
        except OSError:
            pass


def godunov_same(got, ref, tag=None, rel=1e-13):
    """Godunov kernels vs the oracle: bit equality for the STRICT_FP=1 build of libiamrx.so (FMA contraction off, the oracle's
    evaluation order), otherwise <= rel * max(1, |ref|) -- k_godunov.hip is compiled with -ffp-contract=fast by default
    (iamr_amd/csrc/Makefile), as upstream's GPU builds are."""
    import numpy as np
    if os.environ.get("IAMRX_STRICT_FP") == "1":
        assert np.array_equal(got, ref), (tag, float(np.abs(got - ref).max()))
    else:
        err = float(np.abs(got - ref).max())
        assert err <= rel * max(1.0, float(np.abs(ref).max())), (tag, err)


@pytest.fixture(autouse=True)
def _plm_by_default(request):
    """the ORACLE's reconstruction switch (orc_godunov_set_ppm) is process-wide state that its level drivers set from ns.use_ppm at
    every advance; tests of its raw Godunov entry points expect PLM unless they ask for PPM.  (The product takes the scheme as an
    argument of every call and keeps no such state.)"""
    if "gpu" in request.keywords:
        import orc as _orc
        if getattr(_orc, "_LIB", None) is not None:
            _orc.lib().orc_godunov_set_ppm(0)
    yield
