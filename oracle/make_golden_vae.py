"""TEST INFRASTRUCTURE — generates tests/golden/vae_*.pt by running the REAL reference VAEs (imported from
/root/reference: wan23/modules/vae2_2.py and wan/modules/vae.py) on seeded synthetic weights/inputs.

    python oracle/make_golden_vae.py        # build container only

Weights are regenerated from the seed by yume_amd.synth.make_vae_state_dict (fingerprinted by `weight_checksum`).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, vae as ovae  # noqa: E402
from yume_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def build_reference_vae(cfg, sd):
    mod = ref_import.ref_vae("vae2_2" if cfg["version"] == "2.2" else "vae2_1")
    kw = dict(dim=cfg["dim"], z_dim=cfg["z_dim"], dim_mult=cfg["dim_mult"], num_res_blocks=cfg["num_res_blocks"],
              attn_scales=[], temperal_downsample=cfg["temperal_downsample"], dropout=0.0)
    if cfg["version"] == "2.2":
        kw["dec_dim"] = cfg["dec_dim"]
    m = mod.WanVAE_(**kw).eval().requires_grad_(False)
    m.load_state_dict(sd, strict=True)
    return m


def checksum(sd):
    return sum(float(sd[k].double().abs().sum()) for k in sorted(sd))


def main():
    assert ref_import.available()
    os.makedirs(GOLDEN, exist_ok=True)
    for ver in ("2.2", "2.1"):
        cfg = synth.tiny_vae_cfg(ver)
        seed = 7
        sd = synth.make_vae_state_dict(cfg, seed)
        ref = build_reference_vae(cfg, sd)
        mean, inv = ovae.latent_scale(ver)
        s = 16 if ver == "2.2" else 8
        g = torch.Generator().manual_seed(123)
        z = torch.randn(cfg["z_dim"], 3, 4, 6, generator=g)
        video = torch.rand(3, 10, 4 * s, 6 * s, generator=g) * 2 - 1       # 10 frames: the last one is dropped (1+4+4)
        with torch.no_grad():
            dec = ref.decode(z.unsqueeze(0), [mean, inv])[0].clamp(-1, 1)
            enc = ref.encode(video.unsqueeze(0), [mean, inv])[0]
        name = "vae_" + ver.replace(".", "")
        torch.save(dict(version=ver, cfg=cfg, seed=seed, weight_checksum=checksum(sd), z=z, video=video, dec=dec,
                        enc=enc), os.path.join(GOLDEN, name + ".pt"))
        print(name, "dec", tuple(dec.shape), "enc", tuple(enc.shape), os.path.getsize(os.path.join(GOLDEN, name + ".pt")) / 1e6, "MB")


if __name__ == "__main__":
    main()
