"""cg_mrslam_amd -- MI355X-native hot path of mtlazaro/cg_mrslam.

Only what the path needs: ``csrc/`` (HIP kernels + the C ABI of include/cgmr.h, built into
``libcgmr.so``), a ctypes binding (``_lib``), host-side mirrors of the reference interfaces
(``graph.GraphSLAM.optimize``, ``matcher.ScanMatcher``, ``condensed``) and the synthetic
workload generators (``synth``).  There is no CPU fallback: compute entry points raise
``CgmrError`` when libcgmr.so or a gfx950 device is missing.
"""
from ._lib import CgmrError, Context, load_library, library_path  # noqa: F401

__all__ = ["CgmrError", "Context", "load_library", "library_path"]
