#!/usr/bin/env python3
"""The two attention launches of a 5B block as the engine issues them (self-attention 9460 x 9460 x 24 through the automatic choice with
YUME_ATTN_Q_PRESCALED | YUME_ATTN_KV_PADDED -> attn_fwd8's persistent kernel + the merge pass; cross-attention 9460 x 512 x 24 -> the
4-wave kernel), three times each: the workload of tools/run_pmc_attn_traffic.sh (FETCH_SIZE / WRITE_SIZE passes for bench.py's
`roofline.traffic`). Random data."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops
DEV = "cuda"
L, C, H, LC = 9460, 3072, 24, 512
Lp = (L + 63) // 64 * 64
bf = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
qk = torch.zeros(Lp, 2 * C, dtype=torch.bfloat16, device=DEV)
qk[:L, :C] = (bf(L, C).float() * (2.0 * 0.12753)).to(torch.bfloat16)       # q as the engine hands it over: times scale * log2(e)
qk[:L, C:] = bf(L, C) * 2.0
vt = torch.zeros(C, Lp, dtype=torch.bfloat16, device=DEV)
vt[:, :L] = bf(C, L)
kc, vct = bf(LC, C) * 2.0, bf(C, LC)
oa = torch.empty(L, C, dtype=torch.bfloat16, device=DEV)
for it in range(3):
    ops.attn_fwd(qk[:L, :C], qk[:L, C:], vt, oa, L, L, H, q_prescaled=True, kv_padded=True)
    ops.attn_fwd(qk[:L, :C], kc, vct, oa, L, LC, H, q_prescaled=True, kv_padded=True)
torch.cuda.synchronize()
