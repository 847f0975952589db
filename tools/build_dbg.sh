#!/bin/bash
# diagnostic build: the library with -DRHCCQ_STAMPS in k8_minibatch.hip -> dbg_build/librhccq_dbg.so (tools/stamps2.py)
set -e
cd "$(dirname "$0")/.."
python -c "from roibasedimagecompression_amd import build; build.build(verbose=False)"
mkdir -p dbg_build/o
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DRHCCQ_STAMPS -c roibasedimagecompression_amd/csrc/k8_minibatch.hip -o dbg_build/o/k8_minibatch.o
OBJS=$(ls roibasedimagecompression_amd/build/*.o | grep -v k8_minibatch)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dbg_build/librhccq_dbg.so dbg_build/o/k8_minibatch.o $OBJS
echo built dbg_build/librhccq_dbg.so
