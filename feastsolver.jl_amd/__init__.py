"""feastsolver.jl_amd -- MI355X-native FEAST inner loop behind the call
signatures of spacedome/FEASTSolver.jl (``feast!``, ``gen_feast!``, ``nlfeast!``
and the ``factorizer`` / ``left_divider`` / ``finalize!`` seam).

All arithmetic runs in libfeasthip.so (hand-written HIP for gfx950) through its
C ABI (include/feasthip.h); this package is the thin host-side mirror of the
reference's Julia interface.  There is no CPU fallback: importing works without
a GPU (so the C ABI can be inspected), but every solver call raises
``FeastHipError`` when no gfx950 device is present or the library is missing.
"""
from .contour import (CircularContour, RectangularContour, circular_contour_gauss,
                      circular_contour_trapezoidal, in_contour, rectangular_contour_gauss,
                      rectangular_contour_trapezoidal)
from ._lib import FeastHipError, library_path, load_library
from .api import (FeastHandle, HipLU, comm_unique_id, PolynomialNEP, SplitNEP, contour_estimate_eig, dual_gen_feast, feast, finalize, gen_feast, ldiv, lu, mmread, mmwrite, nlfeast, nlfeast_moments, trim_memory)

__all__ = [
    "CircularContour", "RectangularContour", "circular_contour_gauss", "circular_contour_trapezoidal",
    "in_contour", "rectangular_contour_gauss", "rectangular_contour_trapezoidal", "FeastHipError",
    "library_path", "load_library", "FeastHandle", "HipLU", "PolynomialNEP", "SplitNEP", "contour_estimate_eig", "dual_gen_feast", "feast", "finalize",
    "gen_feast", "ldiv", "lu", "mmread", "mmwrite", "nlfeast", "nlfeast_moments", "trim_memory",
]
