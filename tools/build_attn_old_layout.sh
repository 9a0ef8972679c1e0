#!/bin/bash
# experiment build: the r2-r5 P^T layout (v_permlane32_swap) in both one-wave-per-SIMD attention files -> yume_amd/lib/exp/libyume_hip_a7swap.so
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/yume_amd/lib/exp
for f in attn_fwd7 attn_fwd8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-result -DNDEBUG -fno-slp-vectorize -DA7_SWAPFREE=0 -I $root/include \
      -c $root/yume_amd/csrc/$f.hip -o $root/yume_amd/lib/exp/${f}_a7swap.o &
done
wait
objs=$(ls $root/yume_amd/lib/obj/*.o | grep -v "/attn_fwd7.o" | grep -v "/attn_fwd8.o")
/opt/rocm/bin/hipcc -shared -fPIC -Wl,-Bsymbolic --offload-arch=gfx950 -o $root/yume_amd/lib/exp/libyume_hip_a7swap.so $objs $root/yume_amd/lib/exp/attn_fwd7_a7swap.o $root/yume_amd/lib/exp/attn_fwd8_a7swap.o
echo built $root/yume_amd/lib/exp/libyume_hip_a7swap.so
