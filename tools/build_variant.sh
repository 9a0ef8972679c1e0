#!/bin/bash
# tools/build_variant.sh <tag> <file.hip> [-Dflags...] : an experiment build of libyume_hip — <file.hip> recompiled with extra flags, every other
# object taken from the product build — as yume_amd/lib/exp/libyume_hip_<tag>.so (tools/attn_check --lib / tools/gemm_check --lib compare builds)
set -e
tag=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/yume_amd/lib/exp
extra=""; { [ "$src" = attn_fwd7.hip ] || [ "$src" = attn_fwd8.hip ]; } && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-result -DNDEBUG $extra "$@" -I $root/include \
    -c $root/yume_amd/csrc/$src -o $root/yume_amd/lib/exp/${src%.hip}_$tag.o
objs=$(ls $root/yume_amd/lib/obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC -Wl,-Bsymbolic --offload-arch=gfx950 -o $root/yume_amd/lib/exp/libyume_hip_$tag.so $objs $root/yume_amd/lib/exp/${src%.hip}_$tag.o
echo built $root/yume_amd/lib/exp/libyume_hip_$tag.so
