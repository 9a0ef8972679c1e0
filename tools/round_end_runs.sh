#!/bin/bash
# The measurement set a round ends with (one gpurun call): the whole GPU test suite with durations, the default bench under
# rocprofv3 --kernel-trace --stats, the FETCH / WRITE / busy PMC passes of the two attention launches (counters in their own runs) and the
# full default bench as the driver runs it. Results land in gpurun_out/final/; copy what is to be judged into profiles/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final
cd $R
(time timeout 1150 python -m pytest tests -x -q -m gpu --durations=15) > gpurun_out/final/pytest_gpu.log 2>&1
tail -3 gpurun_out/final/pytest_gpu.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-workloads --no-vae > $R/gpurun_out/final/bench_under_rocprof.log 2>&1
cd $R
tail -1 gpurun_out/final/bench_under_rocprof.log | cut -c1-300
timeout 300 bash tools/run_pmc_attn_traffic.sh gpurun_out/final/pmc_attn_traffic > gpurun_out/final/pmc_attn_traffic.log 2>&1
(time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/final/bench_full.log 2>&1
tail -4 gpurun_out/final/bench_full.log | cut -c1-400
