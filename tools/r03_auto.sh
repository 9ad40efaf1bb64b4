#!/bin/bash
# round-3 check of the automatic pool form: GPU suite, then the bench lines
set -o pipefail
O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py > $O/bench.json 2> $O/bench.err && python - <<'PY'
import json
for f in ("bench",):
    d = json.load(open("gpurun_out/r03m/%s.json" % f))
    print(f, round(d["value"]/1e6,1), "M", d["config"]["workload"][-140:], d["roofline"]["frac"])
    for k, v in d.get("sweep", {}).items() if isinstance(d.get("sweep"), dict) else []:
        print("  ", k, v if not isinstance(v, dict) else v.get("value"))
PY
timeout -k 10 120 python bench.py --no-sweep --no-cpu-baseline --contact-off > $O/contact_off.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/contact_off.json'));print('contact_off',d['value']/1e6)"
timeout -k 10 120 python bench.py --no-sweep --no-cpu-baseline --racket-ground > $O/rg.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/rg.json'));print('rg',d['value']/1e6)"
timeout -k 10 120 python bench.py --no-sweep --no-cpu-baseline --env tennis > $O/tennis.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/tennis.json'));print('tennis',d['value']/1e6)"
