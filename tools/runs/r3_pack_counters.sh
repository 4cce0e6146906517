#!/bin/bash
# SQ / TCC / TCP counters of the lean quantize+pack kernel on three b512 layer shapes (outputs under gpurun_out/r3_pack_counters/)
O=$PWD/gpurun_out/r3_pack_counters; mkdir -p $O
R=$PWD
cat > /tmp/pk_layers.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, bench
from cnn_quantization_amd import ops, _lib as L
dev = torch.device('cuda')
for (C, hw) in ((256, 56), (512, 28), (1024, 14)):
    x = bench.laplace_activation((512, C, hw, hw), 3, dev)
    _, parts = ops.act_qdq_per_channel(x, 4, positive=False, clip='laplace', bit_alloc=True, want_parts=True)
    qp, bits = parts['qp'], parts['diag'][L.DIAG_BITS].contiguous()
    bufs = [torch.empty(ops.packed_capacity(x.shape), dtype=torch.uint8, device=dev) for _ in range(3)]
    xs = [x, x.clone(), x.clone()]
    for r in range(2):
        for xx, b in zip(xs, bufs):
            ops.quantize_packed(xx, qp, bits, out=b)
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc$i -o pmc -- python /tmp/pk_layers.py > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/summarize_counters.py $f 'k_pack_lean' > $O/pmc$i.md 2>&1
  find $O/pmc$i -name "*.csv" -size +5M -delete
done
cd $R
cat $O/pmc*.md
