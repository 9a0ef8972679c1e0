"""umT5 text encoder: the CPU oracle against (a) the committed golden vectors produced by the real
reference T5Encoder and (b), when /root/reference is present, the reference class itself; plus host-side checks of the
drop-in module (parameter names, bucket function, prefix-mask equivalence)."""
import sys

import pytest
import torch

from conftest import ROOT, load_golden

sys.path.insert(0, ROOT)
from oracle import ref_import, t5 as ot5  # noqa: E402
from yume_amd import synth, t5  # noqa: E402


def test_oracle_matches_reference_golden():
    fx = load_golden("t5_tiny")
    sd = synth.make_t5_state_dict(fx["cfg"], fx["seed"])
    out = ot5.encoder_forward(sd, fx["cfg"], fx["ids"], fx["mask"])
    for b, n in enumerate(fx["lens"]):
        assert (out[b, :n] - fx["out"][b, :n]).abs().max() <= 2e-5 * fx["out"][b, :n].abs().max()


def test_padding_tokens_can_be_dropped():
    """The equivalence the device path relies on: the valid rows of a masked, padded run equal a run on the valid tokens only."""
    fx = load_golden("t5_tiny")
    sd = synth.make_t5_state_dict(fx["cfg"], fx["seed"])
    for b, n in enumerate(fx["lens"]):
        alone = ot5.encoder_forward(sd, fx["cfg"], fx["ids"][b:b + 1, :n])
        assert (alone[0] - fx["out"][b, :n]).abs().max() <= 2e-5 * fx["out"][b, :n].abs().max()


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present")
def test_oracle_matches_reference_class_directly():
    cfg = synth.tiny_t5_cfg(dim=128, heads=2, ffn=192, layers=3, vocab=300)
    sd = synth.make_t5_state_dict(cfg, 9)
    mod = ref_import.ref_t5()
    ref = mod.T5Encoder(cfg["vocab"], cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_layers"],
                        cfg["num_buckets"], shared_pos=False, dropout=0.0).eval()
    ref.load_state_dict(sd, strict=True)
    ids = torch.randint(1, cfg["vocab"], (2, 150), generator=torch.Generator().manual_seed(1))
    mask = torch.ones(2, 150, dtype=torch.long)
    mask[1, 101:] = 0
    with torch.no_grad():
        want = ref(ids, mask)
    got = ot5.encoder_forward(sd, cfg, ids, mask)
    assert (got[0] - want[0]).abs().max() <= 2e-5 * want.abs().max()
    assert (got[1, :101] - want[1, :101]).abs().max() <= 2e-5 * want.abs().max()
    # the drop-in module restates the bucket function too
    rel = torch.arange(150).unsqueeze(0) - torch.arange(150).unsqueeze(1)
    mine = t5.T5RelativeEmbedding(32, 2, bidirectional=True).buckets(rel)
    assert torch.equal(mine, ref.blocks[0].pos_embedding._relative_position_bucket(rel))
    assert torch.equal(mine, ot5.rel_buckets(150, 150, 32))


def test_dropin_module_has_the_reference_parameter_names():
    cfg = synth.tiny_t5_cfg()
    m = t5.T5Encoder(cfg["vocab"], cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_layers"],
                     cfg["num_buckets"], shared_pos=cfg["shared_pos"])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == synth.t5_param_shapes(cfg)
    m.load_state_dict(synth.make_t5_state_dict(cfg, 1), strict=True)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, dtype=torch.long))                      # CPU tensor: no fallback


def test_umt5_xxl_encoder_parameter_count():
    n = sum(int(torch.Size(s).numel()) for s in synth.t5_param_shapes(synth.T5_CFG_XXL).values())
    # 256384*4096 token embeddings + 24 * (4 * 4096^2 + 3 * 4096*10240 + 2*4096 + 32*64) + 4096
    assert n == 256384 * 4096 + 24 * (4 * 4096 ** 2 + 3 * 4096 * 10240 + 2 * 4096 + 32 * 64) + 4096


@pytest.mark.gpu
def test_t5_encoder_matches_reference_golden_on_gpu():
    fx = load_golden("t5_tiny")
    cfg = fx["cfg"]
    with torch.device("cuda"):
        m = t5.T5Encoder(cfg["vocab"], cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_layers"],
                         cfg["num_buckets"], shared_pos=cfg["shared_pos"])
    m.load_state_dict(synth.make_t5_state_dict(cfg, fx["seed"]), strict=True)
    m = m.cuda().eval()
    out = m(fx["ids"].cuda(), fx["mask"].cuda()).float().cpu()
    assert out.shape == fx["out"].shape
    for b, n in enumerate(fx["lens"]):
        e = ((out[b, :n].double() - fx["out"][b, :n].double()).norm() / fx["out"][b, :n].double().norm()).item()
        print(f"t5 tiny sample {b} ({n} tokens): rel-L2 {e:.3e}")
        assert e <= 1e-2                       # bf16 GEMMs / bf16 P against the fp32 reference
        assert (out[b, n:] == 0).all()
    wrap = t5.T5EncoderModel(text_len=96, device="cuda", model=m)
    lst = wrap.encode_ids(fx["ids"], fx["mask"])
    assert [u.shape[0] for u in lst] == fx["lens"] and torch.equal(lst[1].float().cpu(), out[1, :fx["lens"][1]])
    with pytest.raises(RuntimeError):
        bad = fx["mask"].clone()
        bad[1, 0] = 0
        m(fx["ids"].cuda(), bad.cuda())
