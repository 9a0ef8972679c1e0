#!/bin/bash
# rocprofv3 PMC passes (counters in their own runs; kernel trace only) over tools/pmc_attn_traffic.py -> <outdir>/summary.csv in the format
# bench.py's pmc_traffic_bytes reads (kernel, avg_us, FETCH_SIZE [KiB], TCC hits / misses, WRITE_SIZE [KiB]).
# usage (on the GPU box): bash tools/run_pmc_attn_traffic.sh <outdir>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/pmc_attn_traffic}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $R/tools/pmc_attn_traffic.py > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run busy SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
python $R/tools/pmc_summary.py $OUT $OUT/summary.csv > $OUT/summary.txt
