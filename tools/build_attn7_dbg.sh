#!/bin/bash
# builds yume_amd/lib/dbg/libyume_hip_<name>.so: the library with attn_fwd7.hip compiled under extra defines (timing experiments)
# usage: tools/build_attn7_dbg.sh name1:"-DATTN7_DBG=2" name2:"-DATTN7_RD=8" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p yume_amd/lib/dbg
one() {
  name=${1%%:*}; defs=${1#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -DNDEBUG $defs -I include \
     -c yume_amd/csrc/attn_fwd7.hip -o yume_amd/lib/dbg/attn_fwd7_$name.o
  objs=$(ls yume_amd/lib/obj/*.o | grep -v attn_fwd7)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o yume_amd/lib/dbg/libyume_hip_$name.so $objs yume_amd/lib/dbg/attn_fwd7_$name.o
  echo built $name
}
for a in "$@"; do one "$a" & done
wait
