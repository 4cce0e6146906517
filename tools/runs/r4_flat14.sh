#!/bin/bash
# round 4: the 14x14 layers through the flat tiles (CNNQ_FLAT_MINCPC=49), with and without the LDS rows, against the row-piece kernel
F='s/\| A=.*Gs= *([0-9]+) wgs= *([0-9]+) \| chain +([0-9.]+) us.*group +([0-9.]+) us +([0-9]+) GB.*mismatches=([0-9]+).*/| Gs \1 wgs \2 group \4 us \5 GB\/s(8B) mismatches \6/'
for cfg in "128 0 0" "49 0 0" "49 8 0" "49 0 16" "49 0 8" "128 0 0" "49 8 0"; do set -- $cfg
  echo "MINCPC=$1 KL=$2 K=$3"
  CNNQ_FLAT_MINCPC=$1 CNNQ_FLAT_KL=$2 CNNQ_GRP_K=$3 python tools/bench_group.py --rounds 1 --reps 20 --shapes 1024x14,512x14,256x14 2>&1 | grep "^C=" | sed -E "$F"
done
