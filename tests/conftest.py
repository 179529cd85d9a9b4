import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# processes the tests spawn (tests/test_gpu_dist.py: one process per rank) start their own library: they inherit the suite's choice of
# keeping the callers' boxes (see the `gpu` fixture) through the environment
os.environ.setdefault("IAMRX_COALESCE", os.environ.get("IAMRX_TEST_COALESCE", "0"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run through the HIP C-ABI)")


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
    # The level objects merge the boxes a rank owns (mf.h: coalesce_layout), which would turn every multi-box case of this suite into a
    # single-box run: the suite keeps the callers' boxes as they are, so that ghost exchanges between boxes, partial tiles and the per-box
    # multigrid paths stay covered; tests/test_gpu_coalesce.py covers the merged mode (the library's default).
    lib.tuning_set("COALESCE", float(os.environ.get("IAMRX_TEST_COALESCE", "0")))      # IAMRX_TEST_COALESCE=1: the whole suite on merged boxes
    lib.tuning_set("CHECK_UNIFORM", 1)        # arrays marked uniform (MultiFab::mark_uniform) are verified wherever a solver takes the mark
    return lib


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
