#!/bin/bash
# The measurement set a round ends with (one gpurun call, ~2 min of box time): the default bench under rocprofv3 --kernel-trace --stats,
# the FETCH / WRITE PMC passes of tools/pmc_probe.py (counters in their own runs), the full default bench (CPU baselines included) and
# the 14B workload. Results land in gpurun_out/final/; copy what is to be judged into profiles/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/bench_under_rocprof.log 2>&1
cd $R
tail -1 gpurun_out/final/bench_under_rocprof.log | cut -c1-300
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; name=$1; shift
  (cd /tmp; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/final/pmc/$name -o $name -- python $R/tools/pmc_probe.py > $R/gpurun_out/final/pmc_$name.log 2>&1)
done
python tools/pmc_summary.py gpurun_out/final/pmc gpurun_out/final/pmc_traffic.csv > gpurun_out/final/pmc_summary.txt 2>&1
timeout 330 python bench.py > gpurun_out/final/bench_full.log 2>&1
tail -1 gpurun_out/final/bench_full.log | cut -c1-400
timeout 160 python bench.py --workload 14b --no-cpu-baseline --no-vae > gpurun_out/final/bench_14b.log 2>&1
tail -1 gpurun_out/final/bench_14b.log | cut -c1-300
