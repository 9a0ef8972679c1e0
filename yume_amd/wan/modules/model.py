"""Drop-in WanModel for the Yume-I2V-14B-540P (Wan2.1-I2V-14B) architecture, MI355X-native.

Same contract as yume_amd/wan23/modules/model.py; differences follow the reference file
wan/modules/model.py: the model is always 'i2v' (:609-610), x and y are concatenated on channels (:765-766),
one scalar timestep per sample (:923-928), CLIP image tokens go through img_emb and a second cross-attention
with k_img/v_img whose output is summed before the o projection (:348-389, :939-941), the FramePack branch is
chosen by `rand_num_img >= 0.4` and by a hard-coded `f_num - 9` (:768,781), and forward returns (tensor, cache) with the
block-residual cache of :985-1000 (cache_sample / cache / return_cache / cache_list) implemented as in the reference.
"""
import torch
import torch.nn as nn

from ...wan23.modules.model import Head, WanAttentionBlock as _Block23, WanModel as _Base, _pyramid_conv

__all__ = ["WanModel"]


class MLPProj(nn.Module):
    """parameter layout of reference wan/modules/model.py:529-541 (LN, Linear, GELU, Linear, LN)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.proj = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, in_dim), nn.GELU(),
                                  nn.Linear(in_dim, out_dim), nn.LayerNorm(out_dim))


class WanAttentionBlock(_Block23):
    """14B block: the reference signature has `cross_attn_type` first and per-sample e [B, 6, C]
    (wan/modules/model.py:398-493); the first 257 context tokens are the CLIP image tokens (WanI2VCrossAttention, :362-366)."""

    def __init__(self, cross_attn_type, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True, cross_attn_norm=False, eps=1e-6):
        super().__init__(dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps, with_img=True)
        self.cross_attn_type = cross_attn_type

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens, rand_num_img=None, ids_keep=None,
                ids_restore=None, mask_token=None, seq_lens1=None, cnt_blocks=None):
        packed = rand_num_img is not None and rand_num_img >= 0.4      # rope_apply's per-token branch (:86-100)
        return super().forward(x, e.unsqueeze(1) if e.dim() == 3 else e, seq_lens, grid_sizes, freqs, context, context_lens,
                               ids_keep, ids_restore, mask_token, flag=packed)

    def _n_img(self, ctx_rows):
        return 257


class WanModel(_Base):
    _family = "wan"

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1),
                 qk_norm=True, cross_attn_norm=True, eps=1e-6):
        assert model_type in ("t2v", "i2v")
        nn.Module.__init__(self)
        model_type = "i2v"                      # the reference forces it (:609-610)
        if tuple(patch_size) != (1, 2, 2):
            raise NotImplementedError("patch_size must be (1, 2, 2)")
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
            WanAttentionBlock("i2v_cross_attn", dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps)
            for _ in range(num_layers)])
        self.head = Head(dim, out_dim, patch_size, eps)
        self._bind_blocks()
        self.img_emb = MLPProj(1280, dim)
        self.init_weights()
        self._engine = None
        # NB the reference attaches patch_embedding_{2x,4x,8x,16x,2x_f} after construction
        # (wan/image2video.py:155-159); attach_pyramid() does the same for stand-alone use.

    def attach_pyramid(self):
        self.patch_embedding_2x = _pyramid_conv(self.patch_embedding, (1, 4, 4))
        self.patch_embedding_4x = _pyramid_conv(self.patch_embedding, (1, 8, 8))
        self.patch_embedding_8x = _pyramid_conv(self.patch_embedding, (1, 16, 16))
        self.patch_embedding_16x = _pyramid_conv(self.patch_embedding, (1, 32, 32))
        self.patch_embedding_2x_f = nn.Conv3d(self.in_dim, self.in_dim, kernel_size=(1, 4, 4), stride=(1, 4, 4))
        return self

    def forward(self, x, t, context, seq_len, clip_fea=None, y=None, rand_num_img=None, enable_mask=False,
                latent_frame_zero=9, cache_sample=False, cache=None, return_cache=False, cache_list=None):
        """reference wan/modules/model.py:723-1013; returns (fp32 [C_out, F', H, W] of sample 0, cache|None)."""
        assert clip_fea is not None and y is not None
        if len(x) != 1 or len(y) != 1:
            raise NotImplementedError("yume_amd 14B WanModel: one sample per call (the reference samplers never batch; CFG runs "
                                      "two calls) — got %d" % len(x))
        if enable_mask:
            raise NotImplementedError("enable_mask (MDT token masking) is a training-time path")
        # block-residual cache (:985-1000): `return_cache` records bf16 (x_out - x_in) of the blocks in cache_list (in block
        # order), a later call with the same cache_list adds `cache[cache_list.index(block)]` instead of running the block
        blk_cache = None
        if cache_sample:
            if cache is None:
                cache = []
            blk_cache = ("record" if return_cache else "replay", list(cache_list), cache)
        u = torch.cat([x[0], y[0]], dim=0)
        packed = rand_num_img is not None and rand_num_img >= 0.4
        out = self.engine.forward_one(u, t.reshape(-1)[:1], context[0], clip_fea=clip_fea[0] if clip_fea.dim() == 3
                                      else clip_fea, packed=packed, lfz=latent_frame_zero,
                                      n_sel=(u.shape[1] - 9) if packed else None, cache=blk_cache)
        return out, (cache if cache_sample and return_cache else None)
