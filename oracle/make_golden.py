"""TEST INFRASTRUCTURE — generates tests/golden/dit_*.pt by running the REAL reference (imported from
/root/reference via oracle/ref_import.py) on seeded synthetic inputs/weights (yume_amd/synth.py).

    python oracle/make_golden.py            # build container only (needs /root/reference)

Each fixture holds: the model config, the weight seed (weights are regenerated from the seed by
synth.make_dit_state_dict and fingerprinted by `weight_checksum`), the inputs, the reference output and
the output of every block (forward hooks). The fixtures are what pins oracle/dit.py and the HIP path on
machines without the reference tree (the GPU box).
"""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from yume_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def weight_checksum(sd):
    s = 0.0
    for k in sorted(sd):
        s += float(sd[k].double().abs().sum())
    return s


def build_reference(family, cfg, sd):
    """Instantiate the reference WanModel and load the synthetic weights (strict)."""
    mod = ref_import.ref_dit(family)
    model = mod.WanModel(**cfg)
    if family == "wan":   # the reference attaches these after construction (wan/image2video.py:155-159)
        C, Cin = cfg["dim"], cfg["in_dim"]
        for name, k in (("_2x", 4), ("_4x", 8), ("_8x", 16), ("_16x", 32)):
            setattr(model, "patch_embedding" + name, nn.Conv3d(Cin, C, (1, k, k), stride=(1, k, k)))
        model.patch_embedding_2x_f = nn.Conv3d(Cin, Cin, (1, 4, 4), stride=(1, 4, 4))
    model.load_state_dict(sd, strict=True)
    return model.eval().requires_grad_(False)


def run_reference(model, family, inp, t, seq_len, lfz, packed):
    blocks_out = []
    hooks = [b.register_forward_hook(lambda m, i, o: blocks_out.append(o[0].clone())) for b in model.blocks]
    with torch.no_grad():
        if family == "wan23":
            out = model([inp["x"]], t=t, context=[inp["context"]], seq_len=seq_len, latent_frame_zero=lfz,
                        flag=packed)[0]
        else:
            out, _ = model([inp["x"]], t=t, context=[inp["context"]], seq_len=seq_len, clip_fea=inp["clip_fea"],
                           y=[inp["y"]], rand_num_img=0.6 if packed else 0.2, latent_frame_zero=lfz)
    for h in hooks:
        h.remove()
    return out, blocks_out


CASES = [
    # name, family, F, H, W, lfz, packed, n_text
    ("dit_wan23_packed_f13", "wan23", 13, 12, 16, 8, True, 20),    # branch 1 (history 5)
    ("dit_wan23_packed_f21", "wan23", 21, 12, 16, 8, True, 20),    # branch 2 (history 13)
    ("dit_wan23_plain_f4", "wan23", 4, 12, 16, 8, False, 20),
    ("dit_wan_packed_f13", "wan", 13, 12, 16, 9, True, 20),
    ("dit_wan_plain_f5", "wan", 5, 12, 16, 9, False, 20),
]


def token_count(family, F, H, W, lfz, packed):
    from yume_amd import framepack
    if packed:
        return framepack.pack_plan(F, H, W, lfz, (F - 9) if family == "wan" else None).seq_len
    return F * (H // 2) * (W // 2)


def main():
    assert ref_import.available(), "needs /root/reference"
    os.makedirs(GOLDEN, exist_ok=True)
    for name, family, F, H, W, lfz, packed, n_text in CASES:
        cfg = synth.tiny_cfg(family)
        seed = 11
        sd = synth.make_dit_state_dict(cfg, family, seed)
        model = build_reference(family, cfg, sd)
        inp = synth.make_dit_inputs(cfg, family, F, H, W, n_text=n_text, seed=5)
        L = token_count(family, F, H, W, lfz, packed)
        if family == "wan23" and packed:
            from yume_amd import framepack
            plan = framepack.pack_plan(F, H, W, lfz)
            t = torch.cat([torch.zeros(plan.n_hist_tok, dtype=torch.float64),
                           torch.full((plan.n_new_tok,), 0.731 * 1000, dtype=torch.float64)]).unsqueeze(0)
        else:
            t = torch.tensor([612.5])
        out, blocks_out = run_reference(model, family, inp, t, L, lfz, packed)
        fx = dict(name=name, family=family, cfg=cfg, seed=seed, weight_checksum=weight_checksum(sd), inputs=inp, t=t,
                  seq_len=L, lfz=lfz, packed=packed, out=out,
                  block_rows_stride=4, blocks_out=[b[::4].clone() for b in blocks_out])
        path = os.path.join(GOLDEN, name + ".pt")
        torch.save(fx, path)
        print(f"{name}: L={L} out {tuple(out.shape)} rms {out.pow(2).mean().sqrt():.4f} -> {os.path.getsize(path)/1e6:.2f} MB")


if __name__ == "__main__":
    main()
