"""Per-layer k_minmax / k_qdq timing inside the exact bench.py workload and launch sequence."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnn_quantization_amd import _lib
lib = _lib.load()
layers = bench.build_workload(512, torch.device('cuda'))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
acc = {}
for rep in range(4):
    recs = []
    for L in layers:
        x, y, N, C, HW = L['x'], L['y'], L['N'], L['C'], L['HW']
        G = lib.cnnq_pc_groups(N, C, HW, 1)
        pmm = torch.empty((G, 2, C), dtype=torch.float32, device=x.device); qp = torch.empty((3, C), device=x.device)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record(); lib.cnnq_pc_minmax(x.data_ptr(), N, C, HW, pmm.data_ptr(), st)
        e[1].record(); lib.cnnq_pc_minmax_params(pmm.data_ptr(), G, C, 4, int(L['half']), qp.data_ptr(), st)
        e[2].record(); lib.cnnq_pc_qdq(x.data_ptr(), y.data_ptr(), N, C, HW, qp.data_ptr(), None, None, 1, st)
        e[3].record(); recs.append((L, e))
    torch.cuda.synchronize()
    if rep == 0: continue
    for L, e in recs:
        k = (L['C'], L['HW'], L['half'])
        a = acc.setdefault(k, [0., 0., 0., 0, L['x'].numel()])
        a[0] += e[0].elapsed_time(e[1]); a[1] += e[1].elapsed_time(e[2]); a[2] += e[2].elapsed_time(e[3]); a[3] += 1
tm = tp = tq = 0.
for k, a in acc.items():
    n = a[4]; m, p, q = a[0] / a[3] * 1e-3, a[1] / a[3] * 1e-3, a[2] / a[3] * 1e-3
    cnt = a[3] // 3
    tm += m * cnt; tp += p * cnt; tq += q * cnt
    print('C=%4d HW=%5d half=%d x%2d: minmax %6.1f us %5.0f | params %4.1f us | qdq %6.1f us %5.0f GB/s' % (k[0], k[1], k[2], cnt, m * 1e6, n * 4 / m / 1e9, p * 1e6, q * 1e6, n * 8 / q / 1e9))
print('per forward: minmax %.2f ms, params %.2f ms, qdq %.2f ms' % (tm * 1e3, tp * 1e3, tq * 1e3))
