#!/bin/bash
# VERDICT r4 #8 — "measure what was rejected on paper": fabric reads (FETCH_SIZE) and duration of the ffn.0 GEMM of the 5B block
# (9460 x 14336 x 3072, GELU epilogue; automatic variant) under different tile rasters of the XCD chunk. YUME_GEMM_GROUPM = M-tiles per
# traversal group: an XCD's 32 resident workgroups then cover gm x (32 / gm) tiles — gm = 32 / 37 is the N-band raster (one / two W
# column panels resident per XCD while the A row panels stream), gm = 4 / 8 the near-square patches (the product default is 8), gm = 2 / 16
# the 2 x 16 and 16 x 2 bands. One rocprofv3 process per setting (the knob is read once per process), counters in their own run.
# usage (on the GPU box): bash tools/run_pmc_raster.sh <outdir> [shape]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; SHAPE=${2:-ffn0}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp
for gm in 2 4 8 16 32 37; do
  YUME_GEMM_GROUPM=$gm timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/gm$gm/fetch -o fetch -- python $R/tools/pmc_gemm.py $SHAPE > $OUT/gm$gm.log 2>&1
  python $R/tools/pmc_summary.py $OUT/gm$gm $OUT/gm$gm.csv > $OUT/gm$gm.txt 2>&1
done
for gm in 2 4 8 16 32 37; do echo "== YUME_GEMM_GROUPM=$gm"; cat $OUT/gm$gm.txt; done > $OUT/summary.txt
