#!/usr/bin/env python3
"""Round 6: the b512 step of config 2 with its OUTPUT buffers (and then its inputs) chosen by measured launch time among the
n + M candidate allocations of each size class (n tensors of that shape in the step; tools/place_matrix.py / place_addr.py: the time
of a launch depends on which physical pages its output got - up to 15 % - not on offsets inside an allocation).  Prints the
step before / after.  Development aid; the product form is cnn_quantization_amd/placement.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cnn_quantization_amd import ops  # noqa: E402

dev = torch.device('cuda')
M = int(os.environ.get('M', '8'))
batch = int(os.environ.get('BATCH', '512'))
layers = bench.build_workload(batch, dev)
elems = sum(L['x'].numel() for L in layers)


def step():
    bench.run_step(ops, layers, None)


def t_launch(L, x, y, reps=3):
    ops.act_qdq_per_channel(x, 4, positive=L['half'], out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.act_qdq_per_channel(x, 4, positive=L['half'], out=y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def report(tag):
    t = min(bench.timed_best(step, reps=5) for _ in range(2))
    print('%-34s %.3f ms per step  %.1f G elem/s  path %.3f of 8 TB/s' % (tag, t * 1e3, elems / t / 1e9, elems * 8 / t / 8e12), flush=True)
    return t


classes = {}
for L in layers:
    classes.setdefault(tuple(L['x'].shape), []).append(L)
t0 = report('as allocated')
rejects = []
for which in os.environ.get('WHICH', 'y,x').split(','):
    for shape, Ls in classes.items():
        ref = Ls[0]
        if which == 'y':
            cands = [L['y'] for L in Ls] + [torch.empty_like(ref['x']) for _ in range(M)]
            ts = [t_launch(ref, ref['x'], c) for c in cands]
        else:
            cands = [L['x'] for L in Ls] + [torch.empty_like(ref['x']) for _ in range(M)]
            ts = []
            for c in cands:
                if c.data_ptr() not in [L['x'].data_ptr() for L in Ls]:
                    c.copy_(ref['x'])
                ts.append(t_launch(ref, c, ref['y']))
        order = sorted(range(len(cands)), key=lambda i: ts[i])
        keep = order[:len(Ls)]
        print('%-22s %s: candidates %s -> kept %s' % (list(shape), which, ' '.join('%.0f' % t for t in ts), ' '.join('%.0f' % ts[i] for i in keep)), flush=True)
        if which == 'y':
            for L, i in zip(Ls, keep):
                L['y'] = cands[i]
        else:
            # an input keeps ITS data: copy each tensor's values into the buffer it is given
            olds = [L['x'] for L in Ls]
            news = [cands[i] for i in keep]
            for L, old, new in zip(Ls, olds, news):
                if new.data_ptr() != old.data_ptr():
                    if any(new.data_ptr() == o.data_ptr() for o in olds):
                        continue                      # a tensor of the class already lives there: leave both where they are
                    new.copy_(old)
                    L['x'] = new
        if os.environ.get('HOLD', '1') == '1':
            rejects += [c for i, c in enumerate(cands) if i not in keep]
        del cands
    report('%s planned (n + %d candidates)' % ('outputs' if which == 'y' else 'outputs and inputs', M))
