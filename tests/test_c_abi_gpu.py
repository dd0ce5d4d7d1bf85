"""The boundary from plain C: tests/c_abi/caller.c includes include/cgmr.h, links cg_mrslam_amd/libcgmr.so and calls
cgmr_gn_optimize (GraphSLAM::optimize, src/slam/graph_slam.h:74), cgmr_match_greedy (CharGrid::greedySearch,
src/matcher/chargrid.h:127-185) and cgmr_condense (CondensedGraphCreator::compute, condensed_graph_creator.h:43-50) -- no
Python, no C++ and no torch in the process.  The -m "not gpu" half checks that the header is C99-clean and the link line of
INTEGRATION.md works; the GPU half runs the binary."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "caller.c")
EXE = os.path.join(ROOT, "tests", "c_abi", "caller")


def build_caller(out=EXE):
    """The link line INTEGRATION.md gives a maintainer (plus an rpath so that the test binary finds the in-tree library)."""
    cc = shutil.which("cc") or shutil.which("gcc")
    assert cc, "no C compiler"
    lib_dir = os.path.join(ROOT, "cg_mrslam_amd")
    cmd = [cc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", out,
           "-L" + lib_dir, "-lcgmr", "-lm", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


def test_header_is_plain_c_and_links(tmp_path):
    """include/cgmr.h compiles as C99 with -pedantic -Werror and the caller links against libcgmr.so (no GPU needed)."""
    if not os.path.exists(os.path.join(ROOT, "cg_mrslam_amd", "libcgmr.so")):
        pytest.skip("libcgmr.so not built (__graft_entry__.build())")
    exe = build_caller(str(tmp_path / "caller"))
    assert os.path.exists(exe)
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libcgmr.so" in needed and "libstdc++" not in needed and "python" not in needed.lower()


@pytest.mark.gpu
def test_c_caller_runs_the_three_entry_points():
    exe = EXE if os.path.exists(EXE) else build_caller()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
    for word in ("gn_optimize:", "match_greedy: 1 result", "condense: 3 star edges"):
        assert word in r.stdout
