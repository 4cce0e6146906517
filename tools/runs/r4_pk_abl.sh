#!/bin/bash
# round 4: what the packed single launch spends on arithmetic (FLAT_ABL=8: Q/DQ arithmetic of the register steps compiled out) and
# on the meeting (FLAT_ABL=2), register tiles only (CNNQ_FLAT_KL=0)
for lib in "" tools/alt/libcnnq_abl8.so tools/alt/libcnnq_abl2.so "" tools/alt/libcnnq_abl8.so tools/alt/libcnnq_abl2.so; do
  echo "lib=${lib:-product}"; CNNQ_FLAT_KL=0 CNNQ_HIP_LIB=$lib python tools/bench_pack_single.py 2>&1 | tail -1
done
