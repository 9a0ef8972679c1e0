"""Drop-in boundary, host side (no GPU): the constructor / loading sequence of the reference pipelines executed against the
drop-in classes, checkpoint streaming, the prompt cleaning of the tokenizer wrapper, the attention launch plan."""
import os
import sys
import types
from copy import deepcopy

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
from yume_amd import synth  # noqa: E402


def _upsample_conv3d_weights(conv_small, size):
    """wan23/textimage2video.py:39-57, verbatim behaviour: build a Conv3d and overwrite .weight.data with the interpolated kernel."""
    new_weight = torch.nn.functional.interpolate(conv_small.weight.data, size=size, mode="trilinear", align_corners=False)
    conv_large = torch.nn.Conv3d(in_channels=16, out_channels=5120, kernel_size=size, stride=size, padding=0)
    conv_large.weight.data = new_weight
    if conv_small.bias is not None:
        conv_large.bias.data = conv_small.bias.data.clone()
    return conv_large


def test_textimage2video_constructor_sequence(tmp_path):
    """wan23/textimage2video.py:129-158 with the import swapped to the drop-in: from_config, the pyramid convs / sideblock /
    mask_token attached by hand, then from_pretrained(checkpoint_dir) and load_state_dict(load_file(...))."""
    from safetensors.torch import load_file
    from yume_amd.wan23.modules.model import WanAttentionBlock, WanModel
    tiny = synth.tiny_cfg("wan23")
    src = WanModel(**tiny)
    synth.randomize_module_(src, seed=7)
    src.save_pretrained(tmp_path)
    assert sorted(os.listdir(tmp_path)) == ["config.json", "diffusion_pytorch_model.safetensors"]

    config_wan = {"_class_name": "WanModel", "_diffusers_version": "0.33.0", **{k: v for k, v in tiny.items()}}
    model = WanModel.from_config(config_wan)
    model.patch_embedding_2x = _upsample_conv3d_weights(deepcopy(model.patch_embedding), (1, 4, 4))
    model.patch_embedding_4x = _upsample_conv3d_weights(deepcopy(model.patch_embedding), (1, 8, 8))
    model.patch_embedding_8x = _upsample_conv3d_weights(deepcopy(model.patch_embedding), (1, 16, 16))
    model.patch_embedding_16x = _upsample_conv3d_weights(deepcopy(model.patch_embedding), (1, 32, 32))
    model.patch_embedding_2x_f = torch.nn.Conv3d(48, 48, kernel_size=(1, 4, 4), stride=(1, 4, 4))
    model.sideblock = WanAttentionBlock(model.dim, model.ffn_dim, model.num_heads, model.window_size, model.qk_norm,
                                        model.cross_attn_norm, model.eps)
    model.mask_token = torch.nn.Parameter(torch.zeros(1, 1, model.dim, device=model.device))
    model = WanModel.from_pretrained(str(tmp_path))
    state_dict = load_file(str(tmp_path) + "/diffusion_pytorch_model.safetensors")
    model.load_state_dict(state_dict)
    for k, v in src.state_dict().items():
        assert torch.equal(model.state_dict()[k], v), k
    assert not model.training and model.blocks[1]._owner() is model and model.blocks[1]._index == 1
    # the seam refuses a block nobody owns, loudly (sideblock above is such a block)
    lone = WanAttentionBlock(tiny["dim"], tiny["ffn_dim"], tiny["num_heads"])
    with pytest.raises(RuntimeError, match="owns the block"):
        lone(torch.zeros(1, 4, tiny["dim"]), torch.zeros(1, 4, 6, tiny["dim"]), None, None, None, torch.zeros(1, 2, tiny["dim"]), None)


def test_from_pretrained_streams_shards_and_reports_bad_keys(tmp_path):
    from yume_amd import checkpoint
    from yume_amd.wan.modules.model import WanModel
    cfg = synth.tiny_cfg("wan")
    src = WanModel(**cfg).attach_pyramid()
    synth.randomize_module_(src, seed=8)
    src.save_pretrained(tmp_path, max_shard_size=2 << 20)
    files = checkpoint.weight_files(str(tmp_path))
    assert len(files) > 3 and os.path.exists(tmp_path / "diffusion_pytorch_model.safetensors.index.json")
    # the reference builds the 14B model without the pyramid convs and attaches them afterwards: they are "unexpected" then
    with pytest.raises(RuntimeError, match="unexpected keys"):
        WanModel.from_pretrained(str(tmp_path))
    dst = WanModel.from_config({**cfg, "_class_name": "WanModel"}).attach_pyramid()
    missing, unexpected = checkpoint.stream_state_dict(dst, str(tmp_path))
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # dtype follows the destination parameter (bf16 model, fp32 file)
    dst16 = WanModel.from_config(cfg).attach_pyramid().to(torch.bfloat16)
    checkpoint.stream_state_dict(dst16, str(tmp_path))
    assert dst16.blocks[0].ffn[0].weight.dtype == torch.bfloat16
    assert torch.equal(dst16.blocks[0].ffn[0].weight, src.blocks[0].ffn[0].weight.to(torch.bfloat16))
    # a truncated checkpoint is an error, not a silent partial load
    os.remove(tmp_path / files[0])
    with pytest.raises(Exception):
        checkpoint.stream_state_dict(dst, str(tmp_path))


def test_prompt_cleaning_matches_reference_wrapper(monkeypatch):
    """wan/modules/tokenizers.py:12-22,75-77 (clean='whitespace'). ftfy is not installed in this image: its absence must be loud; with a
    stand-in module the html / whitespace steps are checked against the reference's own functions when the tree is here."""
    from yume_amd import t5
    monkeypatch.setitem(sys.modules, "ftfy", None)
    with pytest.raises(RuntimeError, match="ftfy"):
        t5.clean_prompt("a")
    fake = types.ModuleType("ftfy")
    fake.fix_text = lambda s: s.replace("â€™", "'")
    monkeypatch.setitem(sys.modules, "ftfy", fake)
    raw = "  A &amp;amp; B\n\n walks   through\tthe cityâ€™s  &lt;gate&gt;  "
    assert t5.clean_prompt(raw) == "A & B walks through the city's <gate>"
    ref_file = "/root/reference/wan/modules/tokenizers.py"
    if os.path.exists(ref_file):
        src = open(ref_file).read().split("class HuggingfaceTokenizer")[0].replace("from transformers import AutoTokenizer", "")
        ns = {}
        exec(compile(src, ref_file, "exec"), ns)               # basic_clean / whitespace_clean of the reference, with the stand-in ftfy
        assert ns["whitespace_clean"](ns["basic_clean"](raw)) == t5.clean_prompt(raw)


def test_attention_plan_workspace_sizes():
    """yume_attn_workspace_bytes reflects the launch plan of the one-wave-per-SIMD kernel (host logic, no launch)."""
    from yume_amd import _lib
    lib = _lib.load()
    H = 24
    # 5B shape: 3 heads x 37 query blocks per XCD = 3 rounds + 15 -> the last 5 blocks of every head as 2 key ranges
    rows = 9460 - 32 * 256
    assert lib.yume_attn_workspace_bytes(9460, 9460, H) == 2 * rows * (H * 128 + 2 * H) * 4
    assert lib.yume_attn_workspace_bytes(8192, 9460, H) == 0          # 96 blocks per XCD: whole rounds, nothing to cut
    assert lib.yume_attn_workspace_bytes(9460, 512, H) == 0           # cross-attention: other kernel
    assert lib.yume_attn_workspace_bytes(100, 9460, H) == 0


def test_level0_patch_grid_refuses_odd_sizes():
    from yume_amd import framepack
    with pytest.raises(ValueError, match="must be even"):
        framepack.pack_plan(13, 45, 80, 8)
    assert framepack.pack_plan(13, 44, 80, 8).seq_len == 9460


@pytest.mark.parametrize("family", ["wan23", "wan"])
def test_model_pickles_and_rebinds_its_blocks(family):
    """torch.save(model) / spawn workers pickle the module: the blocks' owner reference is not part of the state (a weakref cannot be
    pickled) and the unpickled model re-binds its blocks; the device engine never travels."""
    import io
    import pickle
    if family == "wan23":
        from yume_amd.wan23.modules.model import WanModel
    else:
        from yume_amd.wan.modules.model import WanModel
    m = WanModel(**synth.tiny_cfg(family))
    synth.randomize_module_(m, seed=3)
    m2 = pickle.loads(pickle.dumps(m))
    assert m2._engine is None and all(b._owner() is m2 and b._index == i for i, b in enumerate(m2.blocks))
    for (k, a), (k2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k == k2 and torch.equal(a, b)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3.blocks[-1]._owner() is m3 and hasattr(m3, "_forward_hooks")
    assert deepcopy(m).blocks[0]._owner() is not m
    assert pickle.loads(pickle.dumps(m.blocks[0]))._owner is None      # a block on its own has no owner


def test_clean_prompt_matches_the_reference_cleaners():
    """yume_amd.t5.clean_prompt against the reference's own basic_clean + whitespace_clean (wan/modules/tokenizers.py:12-22), executed
    from the reference file with ftfy.fix_text stubbed to the identity on BOTH sides (ftfy is not in this image; requirements.txt
    declares it) — the html / whitespace arithmetic is what is compared."""
    ref = "/root/reference/wan/modules/tokenizers.py"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present")
    from yume_amd.t5 import clean_prompt
    src = open(ref).read()
    ns = {}
    stub = types.ModuleType("ftfy")
    stub.fix_text = lambda t: t
    saved = sys.modules.get("ftfy")
    sys.modules["ftfy"] = stub
    try:
        head = src.split("class HuggingfaceTokenizer")[0].replace("from transformers import AutoTokenizer", "")
        exec(compile(head, ref, "exec"), ns)
    finally:
        if saved is None:
            del sys.modules["ftfy"]
        else:
            sys.modules["ftfy"] = saved
    for t in ["  a  cat &amp;amp; a\tdog \n\n walk  ", "&lt;b&gt;bold&lt;/b&gt;", "", "  ", "x", "café   au   lait", "a&nbsp;b"]:
        assert clean_prompt(t, fix_text=lambda s: s) == ns["whitespace_clean"](ns["basic_clean"](t)), repr(t)
    try:
        import ftfy  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="ftfy"):
            clean_prompt("x")
