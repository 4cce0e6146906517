#!/bin/bash
# smoke(), the whole -m gpu suite, one bench.py run and a digest of its JSON line (every step under its own timeout)
O=gpurun_out/full_check; mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; tail -n 4 $O/pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; tail -n 2 $O/bench_full.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/full_check/bench_full.json'))
r=d['roofline']
print(d['ms_per_step'], d['value']/1e9, d['path_frac_hbm_peak'], d['verified'], r['kernel'][:14], r['frac'], r['traffic'], 'class time per step', r['time_per_step_ms'], 'all classes', sum(o['time_per_step_ms'] for o in [r]+list(d['roofline_other_kernels'].values())))
print({k:(round(v["ms"],3), round(v["roofline"]["frac"],3), v["verified"]) for k,v in d["other_configs"].items()}); print("group_status", d["group_status"])
c=d['cpu_baseline']; print(c['value']/1e6, c['cores'], c['one_thread']/1e6, c['by_threads'], c['cpu'], c['config1']['value']/1e6)
PY
