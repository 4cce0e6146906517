"""Round 5, the one counter experiment VERDICT r4 asked for (T1): do the slow buffer placements load the HBM channels unevenly?
Six [512,256,56,56] input / output pairs allocated back to back; each pair runs the single launch (k_mmq_flat) four times in a
row.  Run plain it prints the time per pair; run under `rocprofv3 --kernel-trace --pmc ...` the dispatch order is the key
between the trace's durations and the per-dispatch counter records (tools/stack_probe_report.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import ops
dev = torch.device('cuda')
NP, REPS = int(os.environ.get('PAIRS', '6')), 4
xs = [bench.laplace_activation((512, 256, 56, 56), 40 + i, dev) for i in range(NP)]
ys = [torch.empty_like(x) for x in xs]
ops.act_qdq_per_channel(xs[0][:64].contiguous(), 4)          # workspaces exist before the measured launches
torch.cuda.synchronize()
for i in range(NP):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(REPS + 1)]
    e[0].record()
    for r in range(REPS):
        ops.act_qdq_per_channel(xs[i], 4, out=ys[i])
        e[r + 1].record()
    torch.cuda.synchronize()
    ts = [e[r].elapsed_time(e[r + 1]) * 1e3 for r in range(REPS)]
    print('pair %d  x %#014x y %#014x  us per launch: %s' % (i, xs[i].data_ptr(), ys[i].data_ptr(), ' '.join('%.1f' % t for t in ts)), flush=True)
