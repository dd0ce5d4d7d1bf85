python - <<'PY' 2>&1 | tail -5
import sys; sys.path.insert(0,'.')
import bench
from cg_mrslam_amd import Context
ctx = Context(0)
bench.team_leg(ctx)
for i in range(4):
    o = bench.team_leg(ctx); print(o["key_frames"], round(o["seconds"],4), round(o["key_frames_per_s"],1))
PY
