"""Drop-in WanModel for the Yume-5B-720P (Wan2.2-TI2V-5B) architecture, MI355X-native.

Keeps the reference's constructor arguments, attribute names, state_dict keys and the
`WanModel.forward(x, t, context, seq_len, enable_mask, y, latent_frame_zero, input_ids, flag)`
signature (reference: wan23/modules/model.py:369-495,547-865) so `fastvideo/sample/sample_5b.py`,
`webapp_single_gpu.py` and `wan23/textimage2video.py` can instantiate it unchanged. The modules below
only OWN parameters; the arithmetic runs in hand-written HIP kernels through yume_amd.dit.DiTEngine.
There is no PyTorch fallback: on a machine without the built extension the forward raises.
"""
import json
import math
import os
import weakref

import torch
import torch.nn as nn

from ...dit import DiTEngine

__all__ = ["WanModel"]


class _ScaleOnly(nn.Module):
    """parameter holder for WanRMSNorm (reference model.py:121-137): one `weight` of size dim."""

    def __init__(self, dim, eps):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))


class _Affine(nn.Module):
    """parameter holder for the affine WanLayerNorm used as norm3 (reference model.py:140-150)."""

    def __init__(self, dim, eps):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, eps, with_img=False, qk_norm=True):
        super().__init__()
        self.dim, self.num_heads, self.head_dim, self.eps = dim, num_heads, dim // num_heads, eps
        for name in ("q", "k", "v", "o"):
            setattr(self, name, nn.Linear(dim, dim))
        # reference model.py:175-176: `WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()` — no parameters, q / k pass through
        # (DiTEngine then runs the RoPE kernel with its normalisation switched off)
        self.norm_q = _ScaleOnly(dim, eps) if qk_norm else nn.Identity()
        self.norm_k = _ScaleOnly(dim, eps) if qk_norm else nn.Identity()
        if with_img:
            self.k_img = nn.Linear(dim, dim)
            self.v_img = nn.Linear(dim, dim)
            self.norm_k_img = _ScaleOnly(dim, eps) if qk_norm else nn.Identity()


class WanAttentionBlock(nn.Module):
    """Parameter layout of one DiT block (reference model.py:235-270). Its arithmetic lives in DiTEngine._blocks."""

    def __init__(self, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True, cross_attn_norm=False, eps=1e-6,
                 with_img=False):
        super().__init__()
        self.dim, self.ffn_dim, self.num_heads, self.eps = dim, ffn_dim, num_heads, eps
        self.window_size, self.qk_norm, self.cross_attn_norm = window_size, qk_norm, cross_attn_norm
        self.self_attn = _Attention(dim, num_heads, eps, qk_norm=qk_norm)
        self.norm3 = _Affine(dim, eps) if cross_attn_norm else nn.Identity()
        self.cross_attn = _Attention(dim, num_heads, eps, with_img=with_img, qk_norm=qk_norm)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    # set by the owning WanModel (weak reference: the block is a registered submodule of it)
    _owner = None
    _index = -1

    def _bind(self, model, index):
        object.__setattr__(self, "_owner", weakref.ref(model))
        object.__setattr__(self, "_index", index)

    def __getstate__(self):
        # a weakref cannot be pickled (torch.save(model), spawn workers); the owning WanModel re-binds its blocks in
        # __setstate__, a block unpickled on its own has no owner (forward says so)
        st = dict(self.__dict__)
        st.pop("_owner", None)
        return st

    @staticmethod
    def _rope_rows(freqs, grid, n_tokens, flag):
        """(cos, sin) rows [n, 64, 2] fp32 for the tokens that get RoPE. flag=True (FramePack): `freqs` is the per-token complex
        table [L, 1, 64] (rope_apply, model.py:95-104); else the [1024, 64] axis table indexed by the (f, h, w) grid (:52-70)."""
        if flag:
            fr = freqs.reshape(-1, freqs.shape[-1])[:n_tokens]
        else:
            f, h, w = (int(v) for v in grid)
            c = freqs.shape[-1]
            a, b, d = freqs.reshape(-1, c).split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
            fr = torch.cat([a[:f].view(f, 1, 1, -1).expand(f, h, w, -1), b[:h].view(1, h, 1, -1).expand(f, h, w, -1),
                            d[:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, -1)
        return torch.stack([fr.real, fr.imag], dim=-1).to(torch.float32)

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens, ids_keep=None, ids_restore=None,
                mask_token=None, flag=True):
        """One block through the HIP engine with the reference's arguments (model.py:272-316): x [B, L, C]; e fp32
        [B, L1, 6, C] (the time projection, before this block's modulation is added); freqs as `flag` says (see
        _rope_rows); context [B, Lc, C] already embedded. Tokens at positions >= seq_lens[b] are padding: the reference
        attends only the first seq_lens[b] keys and never reads those rows back; they are returned unchanged here."""
        if ids_keep is not None or ids_restore is not None:
            raise NotImplementedError("ids_keep / ids_restore (MDT token masking) is a training-time path")
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("WanAttentionBlock.forward needs the WanModel that owns the block (it runs on that model's engine)")
        assert e.dtype == torch.float32
        outs = []
        for b in range(x.shape[0]):
            n = int(seq_lens[b]) if seq_lens is not None else x.shape[1]
            rope = self._rope_rows(freqs, None if flag else grid_sizes[b], n, flag)
            nc = int(context_lens[b]) if context_lens is not None else context.shape[1]
            eb = e[b]
            if eb.dim() == 3 and eb.shape[0] not in (1, n):
                eb = eb[:n]
            y = owner.engine.block_forward(self._index, x[b, :n], eb, rope[:n], context[b, :nc], n_img=self._n_img(context[b, :nc]))
            outs.append(torch.cat([y.to(x.dtype), x[b, n:]], dim=0) if n < x.shape[1] else y.to(x.dtype))
        return torch.stack(outs)

    def _n_img(self, ctx_rows):
        return 0


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        self.head = nn.Linear(dim, math.prod(patch_size) * out_dim)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)


def _pyramid_conv(base: nn.Conv3d, size):
    """reference model.py:353-366: a (1,k,k)-stride conv whose kernel is the trilinear up-sampling of the base."""
    oc, ic = base.weight.shape[:2]
    big = nn.Conv3d(ic, oc, kernel_size=size, stride=size)
    with torch.no_grad():
        big.weight.copy_(nn.functional.interpolate(base.weight.detach().float(), size=size, mode="trilinear",
                                                    align_corners=False))
        big.bias.copy_(base.bias.detach())
    return big


class WanModel(nn.Module):
    ignore_for_config = ["patch_size", "cross_attn_norm", "qk_norm", "text_dim", "window_size"]
    _no_split_modules = ["WanAttentionBlock"]
    _family = "wan23"

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1),
                 qk_norm=True, cross_attn_norm=True, eps=1e-6):
        super().__init__()
        assert model_type in ("t2v", "i2v", "ti2v")
        if tuple(patch_size) != (1, 2, 2):
            raise NotImplementedError("patch_size must be (1, 2, 2)")
        assert dim % num_heads == 0 and (dim // num_heads) % 2 == 0
        self.config = dict(model_type=model_type, patch_size=tuple(patch_size), text_len=text_len, in_dim=in_dim,
                           dim=dim, ffn_dim=ffn_dim, freq_dim=freq_dim, text_dim=text_dim, out_dim=out_dim,
                           num_heads=num_heads, num_layers=num_layers, window_size=window_size, qk_norm=qk_norm,
                           cross_attn_norm=cross_attn_norm, eps=eps)
        for k, v in self.config.items():
            setattr(self, k, v)
        self.d = dim // num_heads
        self.mask_ratio, self.mask_token = 0.3, None

        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=patch_size, stride=patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"), nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))
        self.blocks = nn.ModuleList([
            WanAttentionBlock(dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps)
            for _ in range(num_layers)])
        self.head = Head(dim, out_dim, patch_size, eps)
        self._bind_blocks()
        self.init_weights()
        # pyramid patch embeddings are created in the constructor, like the reference (model.py:486-494)
        self.patch_embedding_2x = _pyramid_conv(self.patch_embedding, (1, 4, 4))
        self.patch_embedding_4x = _pyramid_conv(self.patch_embedding, (1, 8, 8))
        self.patch_embedding_8x = _pyramid_conv(self.patch_embedding, (1, 16, 16))
        self.patch_embedding_16x = _pyramid_conv(self.patch_embedding, (1, 32, 32))
        self.patch_embedding_2x_f = nn.Conv3d(in_dim, in_dim, kernel_size=(1, 4, 4), stride=(1, 4, 4))
        self._engine = None

    # diffusers-style conveniences the drivers touch
    @property
    def device(self):
        return self.patch_embedding.weight.device

    @property
    def dtype(self):
        return self.patch_embedding.weight.dtype

    def _bind_blocks(self):
        for i, blk in enumerate(self.blocks):
            blk._bind(self, i)

    # ---- diffusers ModelMixin / ConfigMixin surface the drivers use (wan23/textimage2video.py:143-158, wan/text2video.py:84) ----
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"

    @classmethod
    def from_config(cls, config, **kw):
        cfg = {k: v for k, v in dict(config).items() if k in cls.__init__.__code__.co_varnames and not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, torch_dtype=None, device=None, **kw):
        """Build from `<dir>/config.json` and load `<dir>/diffusion_pytorch_model.safetensors` (or its sharded
        `...safetensors.index.json` form, or a `.bin` / `.pth` pickle) — the on-disk layout diffusers' ModelMixin writes and
        the Yume checkpoints ship in. Tensors are streamed shard by shard straight onto `device` (safetensors' lazy
        `safe_open`), never materialising a second full copy on the host (fastvideo/utils/checkpoint.py:285-337 does the
        same per-tensor copy for its FSDP load). Keys missing from the file keep their initial value only for the pyramid
        patch embeddings (derived from the base kernel, as the reference does); anything else missing or unexpected raises."""
        root = os.fspath(pretrained_model_name_or_path)
        if subfolder:
            root = os.path.join(root, subfolder)
        with open(os.path.join(root, cls.config_name)) as fh:
            config = json.load(fh)
        if device is not None:
            with torch.device(device):
                model = cls.from_config(config, **kw)
        else:
            model = cls.from_config(config, **kw)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        from ...checkpoint import stream_state_dict
        missing, unexpected = stream_state_dict(model, root, cls.weights_name)
        derived = [k for k in missing if k.startswith("patch_embedding_")]
        hard = [k for k in missing if k not in derived]
        if hard or unexpected:
            raise RuntimeError(f"{root}: missing keys {hard[:8]}{'...' if len(hard) > 8 else ''}, "
                               f"unexpected keys {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
        if derived and any(k.startswith("patch_embedding_2x.") for k in derived):
            model._rebuild_pyramid()
        return model.eval()

    def save_pretrained(self, save_directory, max_shard_size=10 << 30, safe_serialization=True):
        """config.json + safetensors shards (+ index) in the layout from_pretrained / diffusers read."""
        from ...checkpoint import save_sharded
        os.makedirs(save_directory, exist_ok=True)
        cfg = {"_class_name": type(self).__name__, "_diffusers_version": "0.33.0"}
        cfg.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()})
        with open(os.path.join(save_directory, self.config_name), "w") as fh:
            json.dump(cfg, fh, indent=2, sort_keys=True)
        save_sharded(self.state_dict(), save_directory, self.weights_name, max_shard_size)

    def _rebuild_pyramid(self):
        for name, size in (("patch_embedding_2x", (1, 4, 4)), ("patch_embedding_4x", (1, 8, 8)),
                           ("patch_embedding_8x", (1, 16, 16)), ("patch_embedding_16x", (1, 32, 32))):
            if hasattr(self, name):
                setattr(self, name, _pyramid_conv(self.patch_embedding, size).to(self.patch_embedding.weight.device,
                                                                                 self.patch_embedding.weight.dtype))

    def init_weights(self):
        """same distributions as reference model.py:892-914 (xavier Linear, N(0,.02) embeddings, zero head)."""
        for mod in self.modules():
            if isinstance(mod, nn.Linear):
                nn.init.xavier_uniform_(mod.weight)
                if mod.bias is not None:
                    nn.init.zeros_(mod.bias)
        nn.init.xavier_uniform_(self.patch_embedding.weight.flatten(1))
        for seq in (self.text_embedding, self.time_embedding):
            for mod in seq.modules():
                if isinstance(mod, nn.Linear):
                    nn.init.normal_(mod.weight, std=.02)
        nn.init.zeros_(self.head.head.weight)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = DiTEngine(self, self._family)
            self._bind_blocks()
        return self._engine

    def __getstate__(self):
        # copies / pickles of the module must not drag the engine (device workspaces, packed weights) along
        st = dict(self.__dict__)
        st["_engine"] = None
        return st

    def __setstate__(self, st):
        super().__setstate__(st)          # nn.Module.__setstate__: state + the hook dictionaries older pickles lack
        self._bind_blocks()

    def enable_sequence_parallel(self, group=None):
        """Split every following forward's tokens over the ranks of `group` (Ulysses all-to-all around the
        self-attention; reference wan23/distributed/sequence_parallel.py:65-176). All ranks must call forward with the
        same inputs and all receive the full output. `enable_sequence_parallel(False)` turns it off."""
        from ...ulysses import SequenceParallel
        self.engine.sp = None if group is False else SequenceParallel(group)
        return self

    def forward(self, x, t, context, seq_len, enable_mask=False, y=None, latent_frame_zero=8, input_ids=None,
                flag=True):
        """x: list of [C_in, F, H, W]; t: [1, seq_len] per-token (flag=True) or [B] / [B, seq_len]; context: list of
        [L, text_dim]; returns list of fp32 [C_out, F', H, W] — reference model.py:547-865."""
        if enable_mask:
            raise NotImplementedError("enable_mask (MDT token masking) is a training-time path, not part of the "
                                      "inference hot path this implementation covers")
        if self.model_type == "i2v":
            assert y is not None
        if y is not None:
            x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]
        if flag and len(x) != 1:
            raise NotImplementedError("the FramePack path is single-sample, as in the reference drivers")
        outs = []
        for i, u in enumerate(x):
            if flag:
                ti = t
            else:
                ti = t[i:i + 1] if t.dim() >= 1 and t.shape[0] == len(x) else t
            outs.append(self.engine.forward_one(u, ti, context[i], packed=bool(flag), lfz=latent_frame_zero))
        return outs
