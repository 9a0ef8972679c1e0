#!/bin/bash
# rocprofv3 PMC passes (counters in their own runs, kernel trace only) of the GEMM kernels, one shape per process.
# usage (on the GPU box): bash tools/run_pmc_gemm.sh <outdir> qkv o cross_q cross_o ffn0 ffn2 | "M N K [variant]" ...     (PMC_WITH_LIB=1 adds the vendor library)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp
for shape in "$@"; do
  tag=$(echo $shape | tr ' ' 'x')
  run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag/$name -o $name -- python $R/tools/pmc_gemm.py $shape > $OUT/${tag}_$name.log 2>&1; }
  run fetch FETCH_SIZE
  run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  run grbm GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE
  python $R/tools/pmc_summary.py $OUT/$tag $OUT/$tag.csv > $OUT/$tag.txt 2>&1
done
