"""Deterministic synthetic weights and inputs for tests and bench.py (there are no checkpoints offline).

Every tensor is drawn from its own CPU generator seeded by (seed, crc32(key)), so the same values are
produced regardless of module construction order, on the build container and on the GPU box alike, and
can be loaded into the reference model, the oracle restatement and the HIP model by key.
Scales follow the reference's init_weights() (wan23/modules/model.py:892-914) except where that would
make a code path vanish (zero biases, zero head weight, unit norm weights): those get small random values.
"""
import math
import zlib

import torch

# model configs: wan23/configs/wan_ti2v_5B.py:8-36 , wan/configs/wan_i2v_14B.py:9-35
CFG_5B = dict(model_type="ti2v", patch_size=(1, 2, 2), text_len=512, in_dim=48, dim=3072, ffn_dim=14336, freq_dim=256,
              text_dim=4096, out_dim=48, num_heads=24, num_layers=30, window_size=(-1, -1), qk_norm=True,
              cross_attn_norm=True, eps=1e-6)
CFG_14B = dict(model_type="i2v", patch_size=(1, 2, 2), text_len=512, in_dim=36, dim=5120, ffn_dim=13824, freq_dim=256,
               text_dim=4096, out_dim=16, num_heads=40, num_layers=40, window_size=(-1, -1), qk_norm=True,
               cross_attn_norm=True, eps=1e-6)


def tiny_cfg(family, dim=512, heads=4, ffn=1024, layers=2, text_dim=256, text_len=64):
    """A small config with head_dim 128 (the only head_dim the reference models use)."""
    base = dict(CFG_5B if family == "wan23" else CFG_14B)
    base.update(dim=dim, num_heads=heads, ffn_dim=ffn, num_layers=layers, text_dim=text_dim, text_len=text_len)
    return base


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
    return g


def _uniform(shape, a, g):
    return (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * a


def _normal(shape, std, g):
    return torch.randn(shape, generator=g, dtype=torch.float32) * std


def dit_param_shapes(cfg, family, pyramid=("_2x", "_4x", "_8x", "_16x", "_2x_f")):
    """state_dict key -> shape for WanModel (reference key names: SURVEY Appendix C)."""
    C, Fd, Cin, Co = cfg["dim"], cfg["ffn_dim"], cfg["in_dim"], cfg["out_dim"]
    sh = {}
    sh["patch_embedding.weight"] = (C, Cin, 1, 2, 2)
    sh["patch_embedding.bias"] = (C,)
    ks = {"_2x": 4, "_4x": 8, "_8x": 16, "_16x": 32}
    for p in pyramid:
        if p == "_2x_f":
            sh["patch_embedding_2x_f.weight"] = (Cin, Cin, 1, 4, 4)
            sh["patch_embedding_2x_f.bias"] = (Cin,)
        else:
            sh[f"patch_embedding{p}.weight"] = (C, Cin, 1, ks[p], ks[p])
            sh[f"patch_embedding{p}.bias"] = (C,)
    for i, (a, b) in {"0": (cfg["text_dim"], C), "2": (C, C)}.items():
        sh[f"text_embedding.{i}.weight"] = (b, a)
        sh[f"text_embedding.{i}.bias"] = (b,)
    for i, (a, b) in {"0": (cfg["freq_dim"], C), "2": (C, C)}.items():
        sh[f"time_embedding.{i}.weight"] = (b, a)
        sh[f"time_embedding.{i}.bias"] = (b,)
    sh["time_projection.1.weight"] = (6 * C, C)
    sh["time_projection.1.bias"] = (6 * C,)
    for l in range(cfg["num_layers"]):
        p = f"blocks.{l}."
        sh[p + "modulation"] = (1, 6, C)
        for att in ("self_attn", "cross_attn"):
            for n in ("q", "k", "v", "o"):
                sh[p + f"{att}.{n}.weight"] = (C, C)
                sh[p + f"{att}.{n}.bias"] = (C,)
            if cfg.get("qk_norm", True):       # (qk_norm=False: nn.Identity, no parameters — reference wan23/modules/model.py:175-176)
                sh[p + f"{att}.norm_q.weight"] = (C,)
                sh[p + f"{att}.norm_k.weight"] = (C,)
        if family == "wan":
            for n in ("k_img", "v_img"):
                sh[p + f"cross_attn.{n}.weight"] = (C, C)
                sh[p + f"cross_attn.{n}.bias"] = (C,)
            if cfg.get("qk_norm", True):
                sh[p + "cross_attn.norm_k_img.weight"] = (C,)
        if cfg["cross_attn_norm"]:
            sh[p + "norm3.weight"] = (C,)
            sh[p + "norm3.bias"] = (C,)
        sh[p + "ffn.0.weight"] = (Fd, C)
        sh[p + "ffn.0.bias"] = (Fd,)
        sh[p + "ffn.2.weight"] = (C, Fd)
        sh[p + "ffn.2.bias"] = (C,)
    sh["head.modulation"] = (1, 2, C)
    sh["head.head.weight"] = (4 * Co, C)
    sh["head.head.bias"] = (4 * Co,)
    if family == "wan":
        sh["img_emb.proj.0.weight"] = (1280,)
        sh["img_emb.proj.0.bias"] = (1280,)
        sh["img_emb.proj.1.weight"] = (1280, 1280)
        sh["img_emb.proj.1.bias"] = (1280,)
        sh["img_emb.proj.3.weight"] = (C, 1280)
        sh["img_emb.proj.3.bias"] = (C,)
        sh["img_emb.proj.4.weight"] = (C,)
        sh["img_emb.proj.4.bias"] = (C,)
    return sh


def make_tensor(key, shape, seed, dim):
    g = _gen(seed, key)
    leaf = key.split(".")[-1]
    if leaf == "modulation":
        return _normal(shape, 1.0 / math.sqrt(dim), g)
    if "norm" in key or key.startswith("img_emb.proj.0") or key.startswith("img_emb.proj.4"):
        return (1.0 + _normal(shape, 0.1, g)) if leaf == "weight" else _normal(shape, 0.05, g)
    if leaf == "bias":
        return _normal(shape, 0.02, g)
    if key.startswith(("text_embedding", "time_embedding", "head.head")):
        return _normal(shape, 0.02, g)
    # xavier-uniform on the flattened [out, fan_in] view (Linear and the patch-embed Conv3d alike)
    fan_out = shape[0]
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return _uniform(shape, math.sqrt(6.0 / (fan_in + fan_out)), g)


_M64 = (1 << 64) - 1


def _s64(v):
    """the 64-bit pattern v as the signed value torch's int64 holds."""
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def hashed_uniform(key, shape, seed, a, device="cpu", chunk=None):
    """Uniform(-a, a) fp32 values that depend only on (seed, key, flat index): a splitmix64-style integer hash evaluated with
    torch's int64 elementwise kernels (wrapping multiply, masked shifts), 23 hash bits -> (2u + 1) * 2^-23 - 1 exactly, one
    rounding for the scale. The same integer and float operations run on the host (all cores, cache-sized chunks, in place:
    ~0.6 G values/s on 8 cores — the sequential CPU generator behind make_tensor takes minutes for a 5B / 14B parameter set)
    and on the GPU, so a full-size weight set can be produced independently on both sides, bit for bit
    (tests/test_synth_hash.py pins the values against a numpy uint64 evaluation)."""
    n = 1
    for s in shape:
        n *= s
    dev = torch.device(device)
    if chunk is None:
        chunk = (1 << 18) if dev.type == "cpu" else (1 << 26)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    base = ((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF) * 0xD1B54A32D192ED03
    idx = torch.arange(0, min(chunk, n), dtype=torch.int64, device=dev)
    x, t = torch.empty_like(idx), torch.empty_like(idx)
    for i0 in range(0, n, chunk):
        m = min(n, i0 + chunk) - i0
        xx, tt = x[:m], t[:m]
        torch.mul(idx[:m], _s64(0x9E3779B97F4A7C15), out=xx)
        xx.add_(_s64(base + i0 * 0x9E3779B97F4A7C15))
        for sh, mul in ((30, 0xBF58476D1CE4E5B9), (27, 0x94D049BB133111EB)):
            torch.bitwise_right_shift(xx, sh, out=tt)
            tt.bitwise_and_((1 << (64 - sh)) - 1)              # logical shift: int64's >> is arithmetic
            xx.bitwise_xor_(tt)
            xx.mul_(_s64(mul))
        torch.bitwise_right_shift(xx, 31, out=tt)
        tt.bitwise_and_((1 << 33) - 1)
        xx.bitwise_xor_(tt)
        xx.bitwise_right_shift_(41).bitwise_and_((1 << 23) - 1)
        o = out[i0:i0 + m]
        o.copy_(xx)
        o.mul_(2.0 ** -22).add_(2.0 ** -23 - 1.0).mul_(a)
    return out.view(shape)


def make_tensor_hashed(key, shape, seed, dim, device="cpu"):
    """make_tensor's per-key scale rules; the large xavier-uniform matrices (every Linear / patch-embed weight of the blocks)
    come from hashed_uniform, the small tensors (biases, norm weights, modulation, N(0, .02) embeddings) from make_tensor."""
    leaf = key.split(".")[-1]
    small = (leaf in ("modulation", "bias") or "norm" in key or key.startswith(("img_emb.proj.0", "img_emb.proj.4"))
             or key.startswith(("text_embedding", "time_embedding", "head.head")))
    if small:
        return make_tensor(key, shape, seed, dim).to(device)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return hashed_uniform(key, shape, seed, math.sqrt(6.0 / (fan_in + shape[0])), device)


class HashedDitStateDict:
    """A read-only mapping key -> fp32 tensor over dit_param_shapes(cfg, family) that GENERATES a tensor on every access
    (make_tensor_hashed) and keeps nothing: the CPU oracle walks a 5B / 14B parameter set block by block with one block's
    weights alive at a time (a 14B fp32 state_dict is 55 GB), and the device model is filled tensor by tensor from the same
    rule evaluated on the GPU (fill_module_hashed_)."""

    def __init__(self, cfg, family, seed=0, pyramid=("_2x", "_4x", "_8x", "_16x", "_2x_f"), device="cpu"):
        self.shapes = dit_param_shapes(cfg, family, pyramid)
        self.cfg, self.family, self.seed, self.device = cfg, family, seed, device

    def __getitem__(self, k):
        return make_tensor_hashed(k, self.shapes[k], self.seed, self.cfg["dim"], self.device)

    def __contains__(self, k):
        return k in self.shapes

    def get(self, k, default=None):
        return self[k] if k in self.shapes else default

    def keys(self):
        return self.shapes.keys()


@torch.no_grad()
def fill_module_hashed_(model, cfg, family, seed=0):
    """every parameter of a device-resident WanModel <- HashedDitStateDict value of its key, generated on the parameter's device."""
    dev = next(model.parameters()).device
    sd = HashedDitStateDict(cfg, family, seed, device=dev)
    for key, p in model.named_parameters():
        p.copy_(sd[key].to(p.dtype))
    return model


def make_dit_state_dict(cfg, family, seed=0, pyramid=("_2x", "_4x", "_8x", "_16x", "_2x_f"), dtype=torch.float32,
                        device="cpu"):
    sd = {}
    for k, shape in dit_param_shapes(cfg, family, pyramid).items():
        sd[k] = make_tensor(k, shape, seed, cfg["dim"]).to(device=device, dtype=dtype)
    return sd


def make_dit_inputs(cfg, family, F, H, W, n_text=77, seed=0):
    """N(0,1) latents / text / CLIP embeddings on CPU (fp32). H, W are LATENT sizes."""
    g = _gen(seed, "inputs")
    out = {"x": torch.randn((cfg["out_dim"] if family == "wan" else cfg["in_dim"], F, H, W), generator=g),
           "context": torch.randn((n_text, cfg["text_dim"]), generator=g)}
    if family == "wan":
        out["y"] = torch.randn((cfg["in_dim"] - cfg["out_dim"], F, H, W), generator=g)
        out["clip_fea"] = torch.randn((1, 257, 1280), generator=g)
    return out


def sampling_sigmas(steps, shift):
    """fastvideo/sample/sample_5b.py:502-506 get_sampling_sigmas."""
    s = torch.linspace(1, 0, steps + 1, dtype=torch.float64)[:steps]
    return (shift * s / (1 + (shift - 1) * s)).tolist()


@torch.no_grad()
def randomize_module_(model, seed=0):
    """Fill every parameter of a (device-resident) WanModel in place with the same per-key scale rules as
    make_tensor, using the parameter's own device generator (fast path for full-size bench models; values differ
    from the CPU generator's)."""
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev)
    dim = model.dim
    for key, p in model.named_parameters():
        g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
        leaf = key.split(".")[-1]
        shape = tuple(p.shape)
        if leaf == "modulation":
            v = torch.randn(shape, generator=g, device=dev) / math.sqrt(dim)
        elif "norm" in key or key.startswith(("img_emb.proj.0", "img_emb.proj.4")):
            v = torch.randn(shape, generator=g, device=dev) * (0.1 if leaf == "weight" else 0.05) + (1.0 if leaf == "weight" else 0.0)
        elif leaf == "bias":
            v = torch.randn(shape, generator=g, device=dev) * 0.02
        elif key.startswith(("text_embedding", "time_embedding", "head.head")):
            v = torch.randn(shape, generator=g, device=dev) * 0.02
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            a = math.sqrt(6.0 / (fan_in + shape[0]))
            v = (torch.rand(shape, generator=g, device=dev) * 2 - 1) * a
        p.copy_(v.to(p.dtype))
    return model


# ----------------------------------------------------------------------------------------------- VAE
# WanVAE_ configurations: wan23/modules/vae2_2.py:748-790,909-1043 (Wan2.2) and wan/modules/vae.py:483-509,591-617 (Wan2.1)
VAE_CFG_22 = dict(version="2.2", dim=160, dec_dim=256, z_dim=48, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
                  temperal_downsample=[False, True, True], patch=2, in_ch=12)
VAE_CFG_21 = dict(version="2.1", dim=96, dec_dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
                  temperal_downsample=[False, True, True], patch=1, in_ch=3)


def tiny_vae_cfg(version, dim=32, dec_dim=None):
    base = dict(VAE_CFG_22 if version == "2.2" else VAE_CFG_21)
    base["dim"] = dim
    base["dec_dim"] = dec_dim if dec_dim is not None else (dim if version == "2.1" else 2 * dim)
    return base


def vae_param_shapes(cfg):
    """state_dict key -> shape of the reference WanVAE_ (both versions), in module order."""
    v22 = cfg["version"] == "2.2"
    z, nres, mult = cfg["z_dim"], cfg["num_res_blocks"], cfg["dim_mult"]
    tds = cfg["temperal_downsample"]
    sh = {}

    def conv(name, cin, cout, k):
        sh[name + ".weight"] = (cout, cin) + tuple(k)
        sh[name + ".bias"] = (cout,)

    def res(name, cin, cout):
        sh[name + ".residual.0.gamma"] = (cin, 1, 1, 1)
        conv(name + ".residual.2", cin, cout, (3, 3, 3))
        sh[name + ".residual.3.gamma"] = (cout, 1, 1, 1)
        conv(name + ".residual.6", cout, cout, (3, 3, 3))
        if cin != cout:
            conv(name + ".shortcut", cin, cout, (1, 1, 1))

    def attn(name, c):
        sh[name + ".norm.gamma"] = (c, 1, 1)
        conv(name + ".to_qkv", c, 3 * c, (1, 1))
        conv(name + ".proj", c, c, (1, 1))

    def resamp(name, c, mode):
        if mode.startswith("up"):
            conv(name + ".resample.1", c, c if v22 else c // 2, (3, 3))
            if mode == "upsample3d":
                conv(name + ".time_conv", c, 2 * c, (3, 1, 1))
        else:
            conv(name + ".resample.1", c, c, (3, 3))
            if mode == "downsample3d":
                conv(name + ".time_conv", c, c, (3, 1, 1))

    # encoder
    dims = [cfg["dim"] * u for u in [1] + mult]
    conv("encoder.conv1", cfg["in_ch"], dims[0], (3, 3, 3))
    li = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        down = i != len(mult) - 1
        mode = "downsample3d" if (tds[i] if i < len(tds) else False) else "downsample2d"
        for j in range(nres):
            res(f"encoder.downsamples.{i}.downsamples.{j}" if v22 else f"encoder.downsamples.{li}", cin, cout)
            cin = cout
            li += 1
        if down:
            resamp(f"encoder.downsamples.{i}.downsamples.{nres}" if v22 else f"encoder.downsamples.{li}", cout, mode)
            li += 1
    res("encoder.middle.0", cout, cout)
    attn("encoder.middle.1", cout)
    res("encoder.middle.2", cout, cout)
    sh["encoder.head.0.gamma"] = (cout, 1, 1, 1)
    conv("encoder.head.2", cout, 2 * z, (3, 3, 3))
    conv("conv1", 2 * z, 2 * z, (1, 1, 1))
    conv("conv2", z, z, (1, 1, 1))
    # decoder
    dims = [cfg["dec_dim"] * u for u in [mult[-1]] + mult[::-1]]
    tus = tds[::-1]
    conv("decoder.conv1", z, dims[0], (3, 3, 3))
    res("decoder.middle.0", dims[0], dims[0])
    attn("decoder.middle.1", dims[0])
    res("decoder.middle.2", dims[0], dims[0])
    li = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        up = i != len(mult) - 1
        mode = "upsample3d" if (tus[i] if i < len(tus) else False) else "upsample2d"
        if not v22 and i in (1, 2, 3):
            cin = cin // 2
        for j in range(nres + 1):
            res(f"decoder.upsamples.{i}.upsamples.{j}" if v22 else f"decoder.upsamples.{li}", cin, cout)
            cin = cout
            li += 1
        if up:
            resamp(f"decoder.upsamples.{i}.upsamples.{nres + 1}" if v22 else f"decoder.upsamples.{li}", cout, mode)
            li += 1
    sh["decoder.head.0.gamma"] = (cout, 1, 1, 1)
    conv("decoder.head.2", cout, cfg["in_ch"], (3, 3, 3))
    return sh


def make_vae_state_dict(cfg, seed=0, device="cpu"):
    sd = {}
    for k, shape in vae_param_shapes(cfg).items():
        g = _gen(seed, k)
        if k.endswith("gamma"):
            v = 1.0 + _normal(shape, 0.1, g)
        elif k.endswith("bias"):
            v = _normal(shape, 0.02, g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            v = _uniform(shape, math.sqrt(3.0 / fan_in), g)
        sd[k] = v.to(device)
    return sd


# ---------------------------------------------------------------------------------------------- umT5 text encoder
T5_CFG_XXL = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32,
                  shared_pos=False)


def tiny_t5_cfg(dim=256, heads=4, ffn=512, layers=2, vocab=1000):
    """umT5 architecture (per-layer relative-position embeddings, head_dim 64) at test size."""
    return dict(vocab=vocab, dim=dim, dim_attn=heads * 64, dim_ffn=ffn, num_heads=heads, num_layers=layers, num_buckets=32,
                shared_pos=False)


def t5_param_shapes(cfg):
    C, Da, F, H = cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"]
    sh = {"token_embedding.weight": (cfg["vocab"], C), "norm.weight": (C,)}
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        sh.update({p + "norm1.weight": (C,), p + "norm2.weight": (C,), p + "attn.q.weight": (Da, C), p + "attn.k.weight": (Da, C),
                   p + "attn.v.weight": (Da, C), p + "attn.o.weight": (C, Da), p + "ffn.gate.0.weight": (F, C),
                   p + "ffn.fc1.weight": (F, C), p + "ffn.fc2.weight": (C, F),
                   p + "pos_embedding.embedding.weight": (cfg["num_buckets"], H)})
    return sh


def make_t5_state_dict(cfg, seed=0, device="cpu"):
    """Deterministic per-key weights: projections ~ fan_in^-1/2 (q additionally / sqrt(head_dim) so the unscaled T5 logits stay
    O(1)), norm weights 1 +- 0.1, relative-position embeddings ~ N(0, 0.5), token embeddings ~ N(0, 1)."""
    sd = {}
    for k, shape in t5_param_shapes(cfg).items():
        g = _gen(seed, k)
        if k.endswith("norm.weight") or "norm1" in k or "norm2" in k:
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif k == "token_embedding.weight":
            v = torch.randn(shape, generator=g)
        elif "pos_embedding" in k:
            v = 0.5 * torch.randn(shape, generator=g)
        else:
            v = torch.randn(shape, generator=g) * shape[1] ** -0.5
            if k.endswith("attn.q.weight"):
                v = v / 8.0
        sd[k] = v.to(device)
    return sd


# ---------------------------------------------------------------------------------------------- CLIP vision tower
CLIP_CFG_VIT_H = dict(image_size=224, patch_size=14, dim=1280, mlp_ratio=4, out_dim=1024, num_heads=16, num_layers=32,
                      pool_type="token", pre_norm=True, post_norm=False, activation="gelu", norm_eps=1e-5)


def tiny_clip_cfg(dim=320, heads=4, layers=3, image=56, patch=14):
    """ViT-H/14 architecture (head_dim 80, pre-norm, nn.GELU, token pooling) at test size: 16 patches + cls."""
    return dict(image_size=image, patch_size=patch, dim=dim, mlp_ratio=4, out_dim=64, num_heads=heads, num_layers=layers,
                pool_type="token", pre_norm=True, post_norm=False, activation="gelu", norm_eps=1e-5)


def clip_param_shapes(cfg):
    C, ps = cfg["dim"], cfg["patch_size"]
    n = (cfg["image_size"] // ps) ** 2 + 1
    M = int(C * cfg["mlp_ratio"])
    sh = {"cls_embedding": (1, 1, C), "pos_embedding": (1, n, C), "head": (C, cfg["out_dim"]), "patch_embedding.weight": (C, 3, ps, ps),
          "pre_norm.weight": (C,), "pre_norm.bias": (C,), "post_norm.weight": (C,), "post_norm.bias": (C,)}
    for i in range(cfg["num_layers"]):
        p = f"transformer.{i}."
        sh.update({p + "norm1.weight": (C,), p + "norm1.bias": (C,), p + "norm2.weight": (C,), p + "norm2.bias": (C,),
                   p + "attn.to_qkv.weight": (3 * C, C), p + "attn.to_qkv.bias": (3 * C,), p + "attn.proj.weight": (C, C),
                   p + "attn.proj.bias": (C,), p + "mlp.0.weight": (M, C), p + "mlp.0.bias": (M,), p + "mlp.2.weight": (C, M),
                   p + "mlp.2.bias": (C,)})
    return sh


def make_clip_state_dict(cfg, seed=0, device="cpu"):
    sd = {}
    for k, shape in clip_param_shapes(cfg).items():
        g = _gen(seed, "clip." + k)
        if "norm" in k:
            v = (1.0 if k.endswith("weight") else 0.0) + 0.1 * torch.randn(shape, generator=g)
        elif k.endswith("bias"):
            v = 0.05 * torch.randn(shape, generator=g)
        elif k in ("cls_embedding", "pos_embedding"):
            v = 0.5 * torch.randn(shape, generator=g)
        elif k == "patch_embedding.weight":
            v = torch.randn(shape, generator=g) * (shape[1] * shape[2] * shape[3]) ** -0.5
        else:
            v = torch.randn(shape, generator=g) * shape[-1 if k == "head" else 1] ** -0.5
        sd[k] = v.to(device)
    return sd
