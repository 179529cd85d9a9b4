"""iamr_amd -- MI355X-native (gfx950) hot path of AMReX-Fluids/IAMR.

The product is libiamrx.so (hand-written HIP kernels + C++ host drivers behind the C-ABI of
include/iamrx.h).  This package is the thin Python loader used by tests and bench.py; it has no
CPU compute path.
"""
from . import lib  # noqa: F401
