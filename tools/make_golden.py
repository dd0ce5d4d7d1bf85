"""Generate the golden fixtures under tests/golden/ for the Gauss-Newton / condensed path.

The expected values come from tests/ref_numpy.py (numpy + SciPy SuperLU + dense inverse), an
implementation independent of both the C oracle and the HIP kernels.  g2o itself is not
available in this image, so these fixtures pin the *restated* algorithm,
not g2o's output: the parity claim for this path stays "unpinned" in DESIGN.md.

Run from the repo root:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cg_mrslam_amd import synth  # noqa: E402
import ref_numpy as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def gn_case(name, V, E, seed, iters, extra_fixed=()):
    g = synth.make_pose_graph(V, E, seed=seed)
    fixed = g["fixed"].copy()
    for v in extra_fixed:
        fixed[v] = 1
    poses, chi2 = R.gn_optimize(g["poses"], fixed, g["edge_from"], g["edge_to"], g["meas"], g["info"], iters)
    query = np.array(sorted(set([1, V // 3, V // 2, V - 1])), dtype=np.int32)
    cov = R.marginals_dense(poses, fixed, g["edge_from"], g["edge_to"], g["meas"], g["info"], query)
    np.savez_compressed(os.path.join(OUT, f"gn_{name}.npz"), poses0=g["poses"], fixed=fixed,
                        edge_from=g["edge_from"], edge_to=g["edge_to"], meas=g["meas"], info=g["info"],
                        iters=iters, poses=poses, chi2=chi2, query=query, cov=cov)
    print(name, "chi2", chi2[0], "->", chi2[-1])


def match_case(name, n_pairs, seed):
    """Matcher golden vectors: inputs + the C oracle's outputs (oracle/matcher_oracle.c).  The reference's own
    matcher cannot be built here (needs Eigen), so these pin the restatement, not the reference binary."""
    from oracle import oracle as O
    sp = synth.make_scan_pairs(n_pairs, seed=seed)
    xyt, score, found = O.close_scan_match_batch(sp["ranges_ref"], sp["ranges_qry"], sp["angle_min"], sp["angle_inc"],
                                                 sp["max_range"], [0, 0, 0], sp["guess"])
    np.savez_compressed(os.path.join(OUT, f"match_{name}.npz"), ranges_ref=sp["ranges_ref"], ranges_qry=sp["ranges_qry"],
                        guess=sp["guess"], true_rel=sp["true_rel"], angle_min=sp["angle_min"], angle_inc=sp["angle_inc"],
                        max_range=sp["max_range"], xyt=xyt, score=score, found=found)
    print(name, "found", int(found.sum()), "/", n_pairs)


def match_case_4096():
    """The 4096-pair subset of C3 that SURVEY.md 8(d) asks for: the first 4096 pairs of the benchmark workload
    (synth.make_scan_pairs(4096, seed=4242)).  The inputs are regenerated from the seed (35 MB of ranges would not be a
    small fixture): the fixture holds their SHA-256 and the C oracle's outputs."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    sp = synth.make_scan_pairs(4096, seed=4242)
    h = hashlib.sha256()
    for k in ("ranges_ref", "ranges_qry", "guess"):
        h.update(np.ascontiguousarray(sp[k]).tobytes())
    nthr = os.cpu_count() or 1
    per = 64

    def run(c):
        lo, hi = c * per, (c + 1) * per
        return O.close_scan_match_batch(sp["ranges_ref"][lo:hi], sp["ranges_qry"][lo:hi], sp["angle_min"], sp["angle_inc"],
                                        sp["max_range"], [0, 0, 0], sp["guess"][lo:hi])
    with ThreadPoolExecutor(max_workers=nthr) as ex:
        parts = list(ex.map(run, range(4096 // per)))
    xyt = np.concatenate([p[0] for p in parts]); score = np.concatenate([p[1] for p in parts]); found = np.concatenate([p[2] for p in parts])
    np.savez_compressed(os.path.join(OUT, "match_close4096.npz"), seed=4242, n_pairs=4096, inputs_sha256=h.hexdigest(),
                        xyt=xyt, score=score, found=found, true_rel=sp["true_rel"])
    print("close4096 found", int(found.sum()), "/ 4096; inputs sha256", h.hexdigest()[:16])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "match4096":
        match_case_4096()
        sys.exit(0)
    gn_case("v60", 60, 110, 11, 6)
    gn_case("v300", 300, 800, 12, 8)
    gn_case("v300_multifix", 300, 800, 13, 8, extra_fixed=(7, 150, 299))
    gn_case("v1200", 1200, 4000, 14, 8)
    match_case("close12", 12, 101)
