"""TEST INFRASTRUCTURE — generates tests/golden/dit_wan_cache.pt: the REAL reference 14B-arch WanModel run with its block-residual
cache (wan/modules/model.py:975-1000): one recording call (cache_sample, return_cache) and one replay call on other inputs.

    python oracle/make_golden_cache.py            # build container only (needs /root/reference)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def reference_cache_run(cfg, sd, a, b, L, lfz, cache_list):
    from make_golden import build_reference
    ref = build_reference("wan", cfg, sd)
    with torch.no_grad():
        out1, cache = ref([a["x"]], t=torch.tensor([700.0]), context=[a["context"]], seq_len=L, clip_fea=a["clip_fea"], y=[a["y"]],
                          rand_num_img=0.6, latent_frame_zero=lfz, cache_sample=True, return_cache=True, cache_list=cache_list)
        out2, none = ref([b["x"]], t=torch.tensor([650.0]), context=[b["context"]], seq_len=L, clip_fea=b["clip_fea"], y=[b["y"]],
                         rand_num_img=0.6, latent_frame_zero=lfz, cache_sample=True, cache=cache, return_cache=False,
                         cache_list=cache_list)
    assert none is None
    return out1, cache, out2


if __name__ == "__main__":
    from test_oracle_dit import _cache_case
    family, cfg, sd, a, b, L, lfz, cache_list = _cache_case()
    out1, cache, out2 = reference_cache_run(cfg, sd, a, b, L, lfz, cache_list)
    path = os.path.join(ROOT, "tests", "golden", "dit_wan_cache.pt")
    torch.save(dict(out_record=out1, cache=[c.clone() for c in cache], out_replay=out2, cache_list=cache_list), path)
    print("wrote", path, [tuple(c.shape) for c in cache])
