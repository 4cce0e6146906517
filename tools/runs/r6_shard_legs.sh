#!/bin/bash
# Round 6: BASELINE configs 3 / 4 / 5 at the 64-sample shard of a batch-sharded run, on one GPU: the forced 1-rank exchange
# through the in-launch route (CNNQ_XRANK=1) and through the collective (CNNQ_XRANK=0), and the same shard without any exchange.
mkdir -p gpurun_out/r6
for x in 1 0; do
  CNNQ_XRANK=$x python bench.py --batch 64 --force-exchange --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r6/bench_b64_force_xr$x.json 2> gpurun_out/r6/bench_b64_force_xr$x.err
done
python bench.py --batch 64 --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r6/bench_b64.json 2> gpurun_out/r6/bench_b64.err
python - <<PY
import json
for f in ("bench_b64_force_xr1", "bench_b64_force_xr0", "bench_b64"):
    d = json.load(open("gpurun_out/r6/%s.json" % f))
    print(f, round(d["ms_per_step"], 3), d["box"], {k: (round(v["ms"], 3), v["verified"], round(v["roofline"]["bytes_moved_per_element"], 2), round(v["roofline"]["frac"], 3)) for k, v in d["other_configs"].items() if k in ("config3", "config4", "config5")})
PY
