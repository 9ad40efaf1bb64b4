#!/bin/bash
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tennis or fuzzed or forced or randomised or maximum_size or scaled" > $O/pytest_tennis.log 2>&1; rc=$?; tail -3 $O/pytest_tennis.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do timeout -k 10 200 python bench.py --env tennis --no-sweep --no-cpu-baseline > $O/t.json 2> $O/t.err && python -c "import json;d=json.load(open('$O/t.json'));print('tennis',d['value']/1e6)"; done
