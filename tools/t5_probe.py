#!/usr/bin/env python3
"""umT5-XXL encoder (random weights, full size: 24 layers, dim 4096, 64 heads, ffn 10240) latency on one MI355X."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import synth, t5

def main():
    dev = "cuda"
    cfg = dict(synth.T5_CFG_XXL)
    with torch.device(dev):
        m = t5.T5Encoder(cfg["vocab"], cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_layers"],
                         cfg["num_buckets"], shared_pos=False)
    m = m.to(torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "norm" in k:
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device=dev))
            elif "pos_embedding" in k:
                p.copy_(0.5 * torch.randn(p.shape, generator=g, device=dev))
            elif k == "token_embedding.weight":
                p.copy_(torch.randn(p.shape, generator=g, device=dev))
            else:
                p.copy_(torch.randn(p.shape, generator=g, device=dev) * p.shape[1] ** -0.5 / (8.0 if k.endswith("attn.q.weight") else 1.0))
    m.eval()
    for n in (77, 256, 512):
        ids = torch.randint(1, cfg["vocab"], (1, 512), device=dev, generator=g)
        mask = torch.zeros(1, 512, dtype=torch.long, device=dev)
        mask[0, :n] = 1
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = m(ids, mask)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        flop = 24 * (2 * n * 4096 * (4 * 4096 + 3 * 10240) + 4 * n * n * 4096)
        print(f"umT5-XXL encode {n} tokens: {dt*1e3:.1f} ms  ({flop/dt/1e12:.1f} TFLOP/s, weights 9.4 GB -> {9.4/dt/1e3:.2f} TB/s)  finite={bool(torch.isfinite(out).all())}", flush=True)
    print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)

if __name__ == "__main__":
    main()
