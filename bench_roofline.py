"""The `roofline` objects of bench.py's JSON line: device time per kernel class, measured live with HIP events on the launch
stream inside a pass that issues exactly what the timed step issues, priced against the kernels' algorithmic bytes and the
8 TB/s HBM3E peak, with the committed rocprofv3 / PMC numbers of `profiles/` next to them.  Split out of bench.py (round 4):
bench.py keeps the driver's contract, this file the measurement behind two of its keys."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_QDQ = 8                  # fused Q/DQ pass, and the resident single launch: 4 B read + 4 B write per element
BYTES_STATS = 4                # statistics pass: one read


def time_kernel_classes(layers, single_launch=True):
    """Device time per kernel class, measured live with HIP events recorded on the launch stream inside ONE pass that
    issues exactly the sequence the product path issues (so cache state is the real one).  On the single-launch routes an
    event is recorded only where the kernel class CHANGES from one tensor to the next (five events per pass: the
    layers come grouped by shape), so the pass runs at the speed of the timed step - with one event per launch boundary
    (round 2) the instrumented pass was 1.5 % slower than the step it explained.  A class's time is the sum of its runs,
    launch gaps inside a run included, exactly as the step pays them.  Returns {class: [seconds, launches, elements]}.
    single_launch=False: the three-launch chain (what runs with several ranks, where the cross-rank exchange sits
    between the statistics and the Q/DQ pass), one event per launch boundary."""
    import ctypes
    from cnn_quantization_amd import _lib, ops
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    resident_ok = single_launch and os.environ.get('CNNQ_RESIDENT', '1') != '0'
    gws = ops._group_workspace(layers[0]['x']) if resident_ok else None
    d = (ctypes.c_int32 * 8)()
    # classify first (no launches), allocate the small tables
    plan = []
    for L in layers:
        N, C, HW = L['N'], L['C'], L['HW']
        group_ok = resident_ok and 0 < lib.cnnq_pc_group_workspace(N, C, HW) <= ops.GROUP_WS_BYTES
        if resident_ok and lib.cnnq_pc_resident_describe(N, C, HW, d) == 0 and not (group_ok and d[6] < 192):
            cls = 'k_mmq_whole'
        elif group_ok:
            cls = 'k_mmq_flat' if (lib.cnnq_pc_group_describe(N, C, HW, d) == 0 and d[2] == 3) else 'k_mmq_group'
        else:
            cls = 'chain'
        G = lib.cnnq_pc_groups(N, C, HW, 1)
        plan.append((L, cls, torch.empty((3, C), dtype=torch.float32, device=L['x'].device),
                     torch.empty((G, 2, C), dtype=torch.float32, device=L['x'].device) if cls == 'chain' else None, G))
    runs, recs = [], []          # single-launch runs: (class, start event, end event, launches, elements); chain records
    cur = None
    for L, cls, qp, pmm, G in plan:
        x, y, N, C, HW = L['x'], L['y'], L['N'], L['C'], L['HW']
        n = x.numel()
        if cls == 'chain':
            if cur is not None:
                cur[2] = torch.cuda.Event(enable_timing=True); cur[2].record(); runs.append(cur); cur = None
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            _lib.check(lib.cnnq_pc_minmax(x.data_ptr(), N, C, HW, pmm.data_ptr(), st), 'minmax')
            e[1].record()
            _lib.check(lib.cnnq_pc_minmax_params(pmm.data_ptr(), G, C, 4, int(L['half']), qp.data_ptr(), st), 'params')
            e[2].record()
            _lib.check(lib.cnnq_pc_qdq(x.data_ptr(), y.data_ptr(), N, C, HW, qp.data_ptr(), None, None, 1, st), 'qdq')
            e[3].record()
            recs.append((n, [('k_minmax', e[0], e[1]), ('k_minmax_params', e[1], e[2]), ('k_qdq', e[2], e[3])]))
            continue
        if cur is None or cur[0] != cls:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            if cur is not None:
                cur[2] = ev
                runs.append(cur)
            cur = [cls, ev, None, 0, 0]
        if cls == 'k_mmq_whole':
            _lib.check(lib.cnnq_pc_minmax_qdq_resident(x.data_ptr(), y.data_ptr(), N, C, HW, 4, int(L['half']),
                                                       qp.data_ptr(), None, st), 'resident')
        else:
            _lib.check(lib.cnnq_pc_minmax_qdq_group(x.data_ptr(), y.data_ptr(), N, C, HW, 4, int(L['half']), gws,
                                                    qp.data_ptr(), None, 0, st), 'group')
        cur[3] += 1
        cur[4] += n
    if cur is not None:
        cur[2] = torch.cuda.Event(enable_timing=True); cur[2].record(); runs.append(cur)
    torch.cuda.synchronize()
    out = {}
    for cls, a, b, launches, elems in runs:
        o = out.setdefault(cls, [0., 0, 0])
        o[0] += a.elapsed_time(b) * 1e-3
        o[1] += launches
        o[2] += elems
    for n, evs in recs:
        for name, a, b in evs:
            o = out.setdefault(name, [0., 0, 0])
            o[0] += a.elapsed_time(b) * 1e-3
            o[1] += 1
            o[2] += n
    return out


KERNEL_BYTES = {'k_qdq': (BYTES_QDQ, 'fused per-channel Q/DQ pass, 8 algorithmic B/elem'),
                'k_minmax': (BYTES_STATS, 'per-channel exact min/max pass, 4 B/elem'),
                'k_mmq_whole': (BYTES_QDQ, 'register-resident min/max + Q/DQ in one launch, whole channels per workgroup, '
                                           '8 algorithmic B/elem'),
                'k_mmq_group': (BYTES_QDQ, 'register-resident min/max + Q/DQ in one launch, extrema exchanged between the '
                                           'workgroups of a channel group (row-piece / whole-channel tiles), 8 algorithmic B/elem'),
                'k_mmq_flat': (BYTES_QDQ, 'register-resident min/max + Q/DQ in one launch, extrema exchanged between the '
                                          'workgroups of a channel, flat tiles of 256*K consecutive float4 of the channel, '
                                          '8 algorithmic B/elem'),
                'k_minmax_params': (0, 'per-channel parameter table (latency-bound, a few KB)')}


def roofline_objects(layers, batch, world, single_launch=True):
    time_kernel_classes(layers, single_launch)        # warm
    kcs = [time_kernel_classes(layers, single_launch) for _ in range(3)]
    objs = {}
    for name in kcs[0]:
        t = min(k[name][0] for k in kcs)
        launches, elems = kcs[0][name][1], kcs[0][name][2]
        by, what = KERNEL_BYTES[name]
        gbs = elems * by / t / 1e9
        objs[name] = {'bound': 'hbm', 'kernel': '%s (%s)' % (name, what), 'achieved': gbs, 'peak': HBM_PEAK_GBS,
                      'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': None, 'launches_per_step': launches,
                      'avg_launch_ms': t * 1e3 / launches, 'bytes_per_launch': elems * by / launches,
                      'time_per_step_ms': t * 1e3}
    # HBM bytes from the PMC counters: collected in separate rocprofv3 --pmc passes of this very command and
    # committed under profiles/ (never measured inside a timed run); attached only to the configuration they
    # were measured on
    pmc = next((f for f in (os.path.join(ROOT, 'profiles', 'r%02d_pmc_traffic.json' % r) for r in (6, 5, 4, 3)) if os.path.exists(f)), '')
    if os.path.exists(pmc):
        try:
            with open(pmc) as f:
                rec = json.load(f)
            for name, o in objs.items():
                k = '%s@b%d' % (name, batch)
                if world == 1 and k in rec.get('bytes_per_launch', {}):
                    o['traffic'] = rec['bytes_per_launch'][k]
                    o['traffic_unit'] = 'bytes per launch'
                    o['traffic_source'] = rec.get('source', 'profiles/' + os.path.basename(pmc))
        except (OSError, ValueError):
            pass
    # the same kernel's average duration under `rocprofv3 --kernel-trace --stats` of this command, committed with the box
    # it was measured on (profiles/rNN_rocprof_headline.json, the newest round's): next to the live figure so the two can be paired
    rp = next((f for f in (os.path.join(ROOT, 'profiles', 'r%02d_rocprof_headline.json' % r) for r in (6, 5, 4)) if os.path.exists(f)), '')
    if os.path.exists(rp) and world == 1:
        try:
            with open(rp) as f:
                rec = json.load(f)
            for name, o in objs.items():
                k = '%s@b%d' % (name, batch)
                if k in rec.get('avg_launch_us', {}):
                    us = rec['avg_launch_us'][k]
                    o['frac_rocprof'] = o['bytes_per_launch'] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
                    o['rocprof_avg_launch_us'] = us
                    o['rocprof_box'] = rec.get('box')
                    o['rocprof_source'] = rec.get('source')
        except (OSError, ValueError):
            pass
    dominant = max((n for n in objs if KERNEL_BYTES[n][0]), key=lambda n: objs[n]['time_per_step_ms'])
    return dominant, objs
