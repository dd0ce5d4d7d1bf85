import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.getcwd())
from cg_mrslam_amd import synth, Context
from cg_mrslam_amd.matcher import ScanMatcher
dev = torch.device("cuda", 0)
ctx = Context(0)
P = 131072
sp = synth.make_scan_pairs(64, seed=4242)
gen = synth.make_scan_pairs_device(P, 990001, dev)
m = ScanMatcher(ctx, sp["n_beams"], sp["angle_min"], sp["angle_inc"], sp["max_range"])
d_xyt = torch.zeros(P, 3, dtype=torch.float64, device=dev); d_score = torch.zeros(P, dtype=torch.float64, device=dev)
d_found = torch.zeros(P, dtype=torch.uint8, device=dev); d_nres = torch.zeros(P, dtype=torch.int32, device=dev)
for rep in range(3):
    for nres in (0, d_nres.data_ptr()):
        m.closeScanMatching_dev(gen["ranges_ref"].data_ptr(), gen["ranges_qry"].data_ptr(), gen["guess"].data_ptr(), P, d_xyt.data_ptr(), d_score.data_ptr(), d_found.data_ptr(), d_nres=nres)
        torch.cuda.synchronize()
        rd = C.c_int64(0); ctx.lib.cgmr_match_last_redo_pairs(ctx.h, C.byref(rd))
        why = (C.c_int64 * 4)(); ctx.lib.cgmr_match_last_path_counts(ctx.h, why)
        st = (C.c_int64 * 2)(); ctx.lib.cgmr_match_last_stats(ctx.h, st)
        print("slow pairs (tiles beyond the LDS pool)", st[1], end="; ")
        print("exhaustive" if nres else "pruned", round(P / m.last_kernel_seconds()), "pairs/s, redo", rd.value, "(borrowed-pool pairs; redo by cause: grid, window/points, lists)", list(why))
