#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03g
mkdir -p $OUT
cd $R || exit 1
for q in 2 3 4; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --no-sweep > $OUT/q${q}_swing.json 2>> $OUT/q.err || exit 1
  GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --no-sweep --contact-off > $OUT/q${q}_swing_off.json 2>> $OUT/q.err || exit 1
  GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --no-sweep --env tennis > $OUT/q${q}_tennis.json 2>> $OUT/q.err || exit 1
  GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --no-sweep --envs-per-gpu 1048576 --rollout-steps 104 --steps 104 > $OUT/q${q}_swing1m.json 2>> $OUT/q.err || exit 1
done
python3 - <<'PY'
import json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r03g")
for q in (2, 3, 4):
    v = [json.load(open(os.path.join(out, "q%d_%s.json" % (q, k)))) for k in ("swing", "swing_off", "tennis", "swing1m")]
    print("GPU_MAX_HW_QUEUES", q, " ".join("%s %.1f M (launch %.2f us)" % (k, d["value"] / 1e6, d["roofline"]["launch_us"]) for k, d in zip(("swing", "contact-off", "tennis", "swing-1M"), v)))
PY
