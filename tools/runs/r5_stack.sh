#!/bin/bash
# round 5: per-channel (TCC instance) write-request counters of k_mmq_flat on fast and slow buffer pairs, one process
O=$PWD/gpurun_out/r5/stack; rm -rf $O; mkdir -p $O
R=$PWD
python -c "import torch;print(torch.cuda.get_device_properties(0))" > $O/box.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $O/counters_list.txt 2>&1
grep -i "TCC_EA0_WRREQ\|TCC_EA0_WR_UNCACHED\|TCC_BUBBLE\|TCC_EA0_WRREQ_STALL\|TCC_EA0_WRREQ_DRAM\|Dimension" $O/counters_list.txt | head -40
timeout 300 python $R/tools/stack_probe.py > $O/plain.log 2>&1; grep pair $O/plain.log
for c in "TCC_EA0_WRREQ" "TCC_EA0_WRREQ_STALL"; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv json -d $O/pmc_$c -o p -- python $R/tools/stack_probe.py > $O/pmc_$c.log 2> $O/pmc_$c.err
  grep pair $O/pmc_$c.log
  ls -la $O/pmc_$c/* | head
done
f=$(find $O/pmc_TCC_EA0_WRREQ -name "*counter_collection.csv" | head -1); head -3 $f; wc -l $f
find $O -name "*.json" -size +20M -delete
