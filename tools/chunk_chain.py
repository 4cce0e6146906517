"""Round 4: does the Infinity Cache serve the Q/DQ pass's read of x when the three-launch chain of config 2 runs CHANNEL
CHUNK by channel chunk (chunk = cc channels x the whole batch, through the strided entry points of the product library)?
One HIP graph per (shape, cc); us per tensor and TB/s on the 8 B/elem accounting, next to the whole-tensor chain and the
single launch.

    python tools/chunk_chain.py [--shapes 256x56,64x112]
"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from cnn_quantization_amd import _lib  # noqa: E402


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='256x56,64x112,512x28,1024x14,256x14')
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--reps', type=int, default=6)
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda')
    ws = ctypes.c_void_p()
    _lib.check(lib.cnnq_group_ws_alloc(32 << 20, ctypes.byref(ws)), 'alloc')
    N = args.batch
    for shp in args.shapes.split(','):
        C, hw = (int(t) for t in shp.split('x'))
        HW = hw * hw
        n = N * C * HW
        nbuf = 3 if n * 4 < (1 << 30) else 2
        xs = [bench.laplace_activation((N, C, hw, hw), 7 + i, dev) for i in range(nbuf)]
        ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
        yref = torch.empty_like(xs[0])
        qp = torch.empty((3, C), dtype=torch.float32, device=dev)
        Gfull = lib.cnnq_pc_groups(N, C, HW, 1)
        pmm_full = torch.empty((Gfull, 2, C), dtype=torch.float32, device=dev)
        s = torch.cuda.Stream()
        st = ctypes.c_void_p(s.cuda_stream)
        with torch.cuda.stream(s):
            _lib.check(lib.cnnq_pc_minmax_qdq(xs[0].data_ptr(), yref.data_ptr(), N, C, HW, 4, 0, pmm_full.data_ptr(), qp.data_ptr(), None, None, st), 'chain')
            s.synchronize()
            it = [0]

            def chain():
                i = it[0] = (it[0] + 1) % nbuf
                lib.cnnq_pc_minmax_qdq(xs[i].data_ptr(), ys[i].data_ptr(), N, C, HW, 4, 0, pmm_full.data_ptr(), qp.data_ptr(), None, None, st)

            def single():
                i = it[0] = (it[0] + 1) % nbuf
                lib.cnnq_pc_minmax_qdq_group(xs[i].data_ptr(), ys[i].data_ptr(), N, C, HW, 4, 0, ws, qp.data_ptr(), None, 0, st)
            t_chain = timed(chain, args.reps)
            t_single = timed(single, args.reps)
            print('[%d,%d,%d,%d] %.0f MB: chain %.1f us (%.2f TB/s 8B), single launch %.1f us (%.2f TB/s)' % (
                N, C, hw, hw, n * 4 / 1e6, t_chain * 1e6, n * 8 / t_chain / 1e12, t_single * 1e6, n * 8 / t_single / 1e12), flush=True)
            for cc in (1, 2, 4, 8, 16, 32, 64, 128):
                if cc >= C:
                    continue
                mb = N * cc * HW * 4 / 2**20
                if mb > 260 or mb < 6:
                    continue
                G = lib.cnnq_pc_groups(N, cc, HW, 1)
                nch = (C + cc - 1) // cc
                pmm = torch.empty((nch, G, 2, cc), dtype=torch.float32, device=dev)
                qps = torch.empty((nch, 3, cc), dtype=torch.float32, device=dev)
                for (lag, rev) in ((0, 0), (0, 1), (1, 0)):
                    graphs = []
                    for i in range(nbuf):
                        x, y = xs[i], ys[i]

                        def enqueue():
                            for k in range(nch + lag):
                                if k < nch:
                                    c0 = k * cc
                                    cn = min(cc, C - c0)
                                    assert cn == cc
                                    _lib.check(lib.cnnq_pc_minmax_strided(x.data_ptr() + c0 * HW * 4, N, cn, HW, C * HW, pmm[k].data_ptr(), st), 'mm')
                                    _lib.check(lib.cnnq_pc_minmax_params(pmm[k].data_ptr(), G, cn, 4, 0, qps[k].data_ptr(), st), 'par')
                                j = k - lag
                                if j >= 0:
                                    c0 = j * cc
                                    _lib.check(lib.cnnq_pc_qdq_strided(x.data_ptr() + c0 * HW * 4, y.data_ptr() + c0 * HW * 4, N, cc, HW, C * HW,
                                                                       qps[j].data_ptr(), None, None, rev, st), 'qdq')
                        enqueue(); s.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=s):
                            enqueue()
                        graphs.append(g)
                    bad = int((ys[0] != yref).sum()) if True else -1

                    def run():
                        i = it[0] = (it[0] + 1) % nbuf
                        graphs[i].replay()
                    t = timed(run, args.reps)
                    print('   chunks of %3d channels = %6.1f MB x %3d, lag %d, reverse %d: %7.1f us  %.2f TB/s (8B)   mismatches vs chain %d' % (
                        cc, mb, nch, lag, rev, t * 1e6, n * 8 / t / 1e12, bad), flush=True)
                    del graphs
        del xs, ys, yref


if __name__ == '__main__':
    main()
