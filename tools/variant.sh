#!/bin/bash
# diagnostic: build a variant of the library with extra -D flags on k8_minibatch.hip (the chain, the mini-batch steps) for same-box A/B runs
# (tools/chain_ab.py):  bash tools/variant.sh NAME -DRHCCQ_G3_KEEP=2 ...   -> dbg_build/librhccq_NAME.so ; prints the chain kernel's registers
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p dbg_build /tmp/rhccq_var
python -c "from roibasedimagecompression_amd import build; build.build(verbose=False)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage \
  -c roibasedimagecompression_amd/csrc/k8_minibatch.hip -o /tmp/rhccq_var/$name.o 2> /tmp/rhccq_var/$name.log || { tail -30 /tmp/rhccq_var/$name.log; exit 1; }
objs=$(ls roibasedimagecompression_amd/build/*.o | grep -v k8_minibatch.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dbg_build/librhccq_$name.so $objs /tmp/rhccq_var/$name.o
grep -A12 "mbk_init3_kernelILi1ELb0ELb0E" /tmp/rhccq_var/$name.log | grep -i "VGPRs:\|Spill\|SGPRs:\|Occupancy" | head -6
