#!/bin/bash
# rocprofv3 PMC passes (counters in their own runs, kernel trace only) of conv_halo_n_kernel on tools/conv_halo_n_probe.py
# usage (on the GPU box): bash tools/run_pmc_halo_n.sh <outdir>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc/$name -o $name -- python $R/tools/conv_halo_n_probe.py > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run grbm GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE
python $R/tools/pmc_summary.py $OUT/pmc $OUT/pmc.csv > $OUT/pmc.txt 2>&1
