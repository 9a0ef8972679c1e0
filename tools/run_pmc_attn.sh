#!/bin/bash
# rocprofv3 PMC passes for one attention problem through tools/attn_check (counters in their own runs; kernel trace only).
# usage (on the GPU box): bash tools/run_pmc_attn.sh <outdir> <variant> [Lq Lk H] [lib] [passes: sq grbm act]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/pmc_attn}; V=${2:-7}; LQ=${3:-8192}; LK=${4:-9460}; H=${5:-24}; LIB=${6:-yume_amd/lib/libyume_hip.so}; PASSES=${7:-sq grbm act}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
case $LIB in /*) ;; *) LIB=$R/$LIB;; esac
mkdir -p $OUT
cd /tmp
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $R/tools/attn_check --lib $LIB --one $LQ $LK $H $V > $OUT/$name.log 2>&1; }
for p in $PASSES; do
  case $p in
    sq) run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;;
    grbm) run grbm GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU;;
    fetch) run fetch FETCH_SIZE;;
    write) run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum;;
    act) run act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM;;
  esac
done
