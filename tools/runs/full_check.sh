#!/bin/bash
O=gpurun_out/full_check; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; tail -n 4 $O/pytest_all.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; tail -n 2 $O/bench_full.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/full_check/bench_full.json'))
print(d['ms_per_step'], d['value']/1e9, d['path_frac_hbm_peak'], d['verified'], d['roofline']['kernel'][:14], d['roofline']['frac'], d['roofline']['traffic'])
print({k:(round(v["ms"],3), round(v["roofline"]["frac"],3), v["verified"]) for k,v in d["other_configs"].items()}); print("group_status", d["group_status"])
c=d['cpu_baseline']; print(c['value']/1e6, c['cores'], c['one_thread']/1e6, c['by_threads'], c['cpu'], c['config1']['value']/1e6)
PY
