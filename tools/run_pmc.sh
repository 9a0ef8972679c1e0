#!/bin/bash
# rocprofv3 PMC passes for the dominant kernels (counters in their own runs; no sys/hip/hsa trace domains).
# usage (on the GPU box): bash tools/run_pmc.sh <outdir>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/pmc}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $R/tools/pmc_probe.py > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run grbm GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU
find $OUT -name "*.csv" | head -20
