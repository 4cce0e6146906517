#!/bin/bash
# round 6: where the packed single launch (config 2 with the 4-bit codes as the stored result, 4.5 B/elem) loses
# its time.  Development builds of the library with parts of k_mmq_flat / k_mmq_group compiled out (timing only, WRONG results):
#   tools/build_alt.sh abl$n -DFLAT_ABL=$n   n: bit 0 no stores, bit 1 no meeting, bit 3 no quotient arithmetic
# against the shipped library, all on one box in one call; and the alternative the reviewer proposed - the chain's two passes
# (read x for the extrema, read it again to quantize + pack: 8.5 B/elem).
mkdir -p gpurun_out/r6
O=gpurun_out/r6/pack_ablation.txt; : > $O
python -c "import bench; print('box', bench.box_id(0))" 2>/dev/null | tee -a $O
echo "shipped library" | tee -a $O
python tools/bench_pack_single.py 2>/dev/null | tee -a $O
for n in 8 2 10 1 3 11; do
  echo "FLAT_ABL=$n" | tee -a $O
  CNNQ_HIP_LIB=$PWD/tools/alt/libcnnq_abl$n.so python tools/bench_pack_single.py 2>/dev/null | tee -a $O
done
echo "the chain: k_minmax + parameters + k_q_pack4 (8.5 B/elem)" | tee -a $O
python - <<'PY' 2>/dev/null | tee -a $O
import torch, bench
from cnn_quantization_amd import ops, _lib as L
dev = torch.device('cuda')
layers = bench.build_workload(512, dev, seed=1)
def step():
    for Ly in layers:
        x = Ly['x']
        st, _ = None, None
        y, parts = None, None
        # the extrema pass + parameter table (chain form), then the quantize + pack pass
        N, C, HW = Ly['N'], Ly['C'], Ly['HW']
        lib = L.load()
        G = lib.cnnq_pc_groups(N, C, HW, 1)
        pmm = torch.empty((G, 2, C), dtype=torch.float32, device=dev)
        qp = torch.empty((L.NQP, C), dtype=torch.float32, device=dev)
        L.check(lib.cnnq_pc_minmax(x.data_ptr(), N, C, HW, pmm.data_ptr(), ops._stream(x)), 'minmax')
        L.check(lib.cnnq_pc_minmax_params(pmm.data_ptr(), G, C, 4, int(Ly['half']), qp.data_ptr(), ops._stream(x)), 'params')
        ops.quantize_pack4(x, qp)
t = bench.timed_best(step, reps=3)
n = sum(Ly['x'].numel() for Ly in layers)
print('chain: %.3f ms per forward, %.1f G elem/s, %.3f of 8 TB/s on the 8.5 B/elem it moves' % (t * 1e3, n / t / 1e9, n * 8.5 / t / 8e12))
PY
