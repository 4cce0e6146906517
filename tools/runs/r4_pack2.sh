python -m pytest tests/test_single_outputs_gpu.py -x -q -k "packed" 2>&1 | tail -2
for v in "" "CNNQ_PK_PLAIN=1" "CNNQ_PK_NARROW=1"; do echo "== $v"; env $v python tools/bench_pack_layers.py 2>&1 | grep -v amdgpu.ids; done
