#!/bin/bash
set -o pipefail
O=gpurun_out/r03p; mkdir -p $O
TB_BENCH_FORCE_COLLECTIVE=1 timeout -k 10 200 python bench.py --no-sweep --no-cpu-baseline 2> $O/forced.err | grep '^{' > $O/forced.json && python - <<'PY'
import json
d = json.load(open("gpurun_out/r03p/forced.json"))
print(d["value"]/1e6, d["config"]["workload"][-420:]); print(json.dumps(d["exchange"])[:1500])
PY
timeout -k 10 600 python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -3
