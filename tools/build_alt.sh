#!/bin/bash
# development builds of the library under tools/alt/ (git-ignored): tools/build_alt.sh <name> <extra hipcc flags...>
# e.g. tools/build_alt.sh abl8 -DFLAT_ABL=8 ; use with CNNQ_HIP_LIB=tools/alt/libcnnq_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/alt
FLAGS=$(python -c "from cnn_quantization_amd import _build; print(' '.join(_build.FLAGS))")
/opt/rocm/bin/hipcc $FLAGS -I include "$@" cnn_quantization_amd/csrc/cnnq_kernels.hip -o tools/alt/libcnnq_$name.so
echo tools/alt/libcnnq_$name.so
