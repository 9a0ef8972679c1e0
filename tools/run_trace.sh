#!/bin/bash
# per-workgroup time stamps of experiment builds (csrc/trace.hpp): tools/build_variant.sh trace_attn attn_fwd7.hip -DYUME_TRACE and
# tools/build_variant.sh trace_gemm gemm_bf16.hip -DYUME_TRACE first. Usage: tools/run_trace.sh [attn] [gemm]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/trace
A=yume_amd/lib/exp/libyume_hip_trace_attn.so
G=yume_amd/lib/exp/libyume_hip_trace_gemm.so
what="${*:-attn gemm}"
if [[ $what == *attn* ]]; then
timeout 120 tools/attn_check --lib $A --one 9460 9460 24 --trace gpurun_out/trace/attn_5b.bin 256
timeout 120 tools/attn_check --lib $A --one 8192 9460 24 --trace gpurun_out/trace/attn_8192.bin 256
timeout 120 tools/attn_check --lib $A --one 9460 512 24 --trace gpurun_out/trace/attn_cross_v7.bin 263
fi
if [[ $what == *gemm* ]]; then
g() { timeout 120 tools/gemm_check --lib $G --variants 0 --reps 3 --one $2 $3 $4 $5 $6 --trace gpurun_out/trace/gemm_$1.bin; }
g o 9460 3072 3072 3 2
g crossq 9460 3072 3072 0 0
g ffn0 9460 14336 3072 1 0
g ffn2 9460 3072 14336 3 2
g 8192 8192 8192 8192 0 0
g qkv 9460 9216 3072 4 0
fi
for f in gpurun_out/trace/*.bin; do echo "== $f"; python tools/trace_report.py $f; done > gpurun_out/trace/report.txt 2>&1
