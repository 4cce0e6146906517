#!/bin/bash
# which dispatch block (CNNQ_GRP_CB) does this box prefer, and what kind of box is it?
python tools/box_class.py 2>&1 | grep "^box"
for r in 1 2; do for cb in 1 2 4; do
  echo "== round $r cb=$cb"; CNNQ_GRP_CB=$cb python tools/bench_group.py --rounds 1 --reps 24 --shapes 64x112,256x56,128x56,512x28,64x56,128x28 2>&1 | grep "^C=" | sed -E 's/\| A=.*wgs= *([0-9]+) \| chain +([0-9.]+) us.*group +([0-9.]+) us +([0-9]+) GB.*/| group \3 us \4 GB\/s(8B)/'
done; done
python tools/box_class.py 2>&1 | grep "^box"
