"""Cycle marks of workgroup 0 of the LAST k_match_greedy launch of one hierarchical search (timing build: CGMR_LIB=.../libcgmr_t.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cg_mrslam_amd import Context, synth
from cg_mrslam_amd._lib import load_library
from tests.test_matcher_gpu import _lc
ctx = Context(0)
sp = synth.make_scan_pairs(2, seed=91)
m = _lc(ctx, sp)
region = np.array([[-10, -5, np.float32(-np.pi), 10, 5, np.float32(np.pi)]], dtype=np.float32)
ref = m.cartesian(sp["ranges_ref"][0]); q = m.subsample(m.cartesian(sp["ranges_qry"][0]))
for levels in (2, 4):
    for _ in range(2): got = m.hierarchicalSearch(ref, q, region, 0.025, 0.2, 0.5, 0.5, 0.2, levels)
    o = np.zeros(16, dtype=np.uint64)
    load_library().cgmr_debug_gphase(C.c_void_p(o.ctypes.data))
    o = o.astype(np.int64)
    names = ["load grid / rasterise", "item header + sincos", "list build", "gather", "sums + bins (to the end)"]
    print(f"levels {levels}: results {len(got)}, ref {len(ref)} query {len(q)} points; cycles of workgroup 0, last launch:",
          {n: int(o[i + 1] - o[i]) for i, n in enumerate(names)}, "total", int(o[5] - o[0]))
