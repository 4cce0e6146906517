#!/bin/bash
# round 5: the quantizers inside a real forward (MIOpen convolutions interleaved): per-layer times of configs 2, 3 (ResNet-50 b512)
# and 5 (VGG-16 b512) from harness/inference_sim.py
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
H="python -m cnn_quantization_amd.harness.inference_sim"
timeout 900 $H -a resnet50 -b 512 -pcq_w -pcq_a --qtype int4 -qw int4 > gpurun_out/r5/r05_harness_resnet50_b512_cfg2.log 2>&1
timeout 900 $H -a resnet50 -b 512 -pcq_w -pcq_a --qtype int4 -qw int4 -c laplace -baa -baw -bcw > gpurun_out/r5/r05_harness_resnet50_b512_cfg3.log 2>&1
timeout 900 $H -a vgg16 -b 512 -pcq_w -pcq_a --qtype int4 -qw int4 -c laplace -baa -baw -mtq -me > gpurun_out/r5/r05_harness_vgg16_b512_cfg5.log 2>&1
for f in gpurun_out/r5/r05_harness_*.log; do echo "== $f"; tail -6 $f; done
