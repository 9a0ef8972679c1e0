export TMPDIR=/tmp
python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r2_bench_v3.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench_v3.log').read())
print(d['value'], d['ms_per_step'], d.get('cached_context_ms_per_step'), d['vae_decode']['latents_per_s'])
for r in [d['roofline']]+d['roofline_all']:
    print(r['group'], round(r.get('launch_ms',0),4), round(r.get('ms_per_step',0),2), round(r.get('achieved',0),1), r.get('traffic'))
print(d.get('parity'))
PY
