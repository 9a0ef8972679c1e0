export TMPDIR=/tmp
timeout 200 tools/gemm_check 2>&1 | grep "epi=3\|failure" | grep -v "big/auto" | cut -c1-64,95-220
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-vae 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/b.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/b.log').read())
print(d['value'], d['ms_per_step'])
for r in [d['roofline']]+d['roofline_all']:
    print(r['group'], round(r.get('launch_ms',0),4), round(r.get('ms_per_step',0),2), round(r.get('achieved',0),1))
PY
