#!/bin/bash
# A/B of environment switches on the default bench (quick form: no CPU legs, no VAE, no side workloads). usage: tools/bench_ab.sh "VAR=a" "VAR=b" ...
export TMPDIR=/tmp
for kv in "$@"; do
  env $kv python bench.py --no-cpu-baseline --no-vae --no-workloads --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > /tmp/ab.json
  python - "$kv" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read())
r = {x['group']: x for x in [d['roofline']] + d['roofline_all']}
print(sys.argv[1], 'ms_per_step', round(d['ms_per_step'], 3), 'trimmed', round(d.get('trimmed_last_block_ms_per_step') or 0, 3), 'cached', round(d.get('cached_context_ms_per_step') or 0, 3),
      'sustained', round((d.get('calibration') or {}).get('mfma_sustained_tflops') or 0), ' '.join(f"{k}={r[k]['launch_ms']*1e3:.1f}" for k in ('attn_self', 'gemm_qkv', 'gemm_ffn0', 'gemm_ffn2', 'gemm_o', 'gemm_cross_o', 'gemm_cross_q', 'attn_cross', 'adaln', 'rmsnorm_rope') if k in r))
PY
done
