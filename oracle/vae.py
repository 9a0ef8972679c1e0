"""TEST INFRASTRUCTURE — CPU restatement (plain torch, fp32) of the reference's causal 3D VAE encode/decode:

  version "2.2" = wan23/modules/vae2_2.py  (WanVAE_: z=48, dim=160, dec_dim=256, patchify 2, Avg/Dup shortcuts)
  version "2.1" = wan/modules/vae.py       (WanVAE_: z=16, dim=96, upsample convs halve the channels)

It is the checker for the HIP VAE path; nothing under yume_amd/ imports it. Functional restatement on a flat
state_dict (reference key names), with the chunked feat_cache semantics made explicit:
  * every CausalConv3d with a temporal kernel keeps the last two INPUT frames of the previous chunk (vae2_2.py:17-44,
    216-239); a cache of one frame is left-padded with a zero frame;
  * upsample3d's time_conv is SKIPPED on the first chunk ('Rep', :116-121) and sees a zero cache on the second;
  * downsample3d's strided time_conv is skipped on the first chunk and then runs on [last frame of previous chunk] + x;
  * encode feeds frames 1, 4, 4, ... (only 1 + 4*((T-1)//4) frames are used, :802-820); decode feeds one latent frame
    at a time (:839-857) and DupUp3D drops its first duplicated frame on the first chunk (:416-417).
Parity of THIS file is pinned against the real reference (tests/test_oracle_vae.py) and the golden vectors in
tests/golden/vae_*.pt generated from the real reference by oracle/make_golden_vae.py.
"""
import torch
import torch.nn.functional as F

CFG_22 = dict(version="2.2", dim=160, dec_dim=256, z_dim=48, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
              temperal_downsample=[False, True, True], patch=2, in_ch=12)
CFG_21 = dict(version="2.1", dim=96, dec_dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
              temperal_downsample=[False, True, True], patch=1, in_ch=3)

# vae2_2.py:925-1033 / vae.py:626-638 latent statistics
MEAN_22 = [-0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891, 0.1570,
           -0.0098, 0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502, -0.2055, -0.0322,
           0.1109, 0.1567, -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723, -0.2839, -0.2083, -0.0520,
           0.3748, 0.0152, 0.1957, 0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667]
STD_22 = [0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901, 0.6885, 0.6165,
          0.8454, 0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659, 0.7069, 0.5338, 0.4889, 0.4917,
          0.4069, 0.4999, 0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944, 0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600,
          0.3943, 0.5537, 0.5444, 0.4089, 0.7468, 0.7744]
MEAN_21 = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922,
           -0.9497, 0.2503, -0.2921]
STD_21 = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253,
          2.8251, 1.9160]


def latent_scale(version):
    m, s = (MEAN_22, STD_22) if version == "2.2" else (MEAN_21, STD_21)
    return torch.tensor(m), 1.0 / torch.tensor(s)


# ------------------------------------------------------------------------------------------- primitives
class Cache:
    """per-conv temporal state across chunks, keyed by the conv's state_dict prefix."""

    def __init__(self):
        self.d = {}
        self.chunk = 0


def causal_conv(sd, name, x, cache):
    """CausalConv3d.forward with feat_cache handling (vae2_2.py:34-44 + the cache bookkeeping around each call,
    e.g. :219-234). x [C,T,H,W] -> [Co,T,H,W]."""
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    kt, kh, kw = w.shape[2:]
    if kt == 1:
        return F.conv3d(F.pad(x.unsqueeze(0), (kw // 2, kw // 2, kh // 2, kh // 2)), w, b)[0]
    prev = cache.d.get(name)
    keep = x[:, -2:]
    if keep.shape[1] < 2 and prev is not None:
        keep = torch.cat([prev[:, -1:], keep], dim=1)
    xin = x if prev is None else torch.cat([prev, x], dim=1)
    pad_t = 2 - (0 if prev is None else prev.shape[1])
    xin = F.pad(xin.unsqueeze(0), (kw // 2, kw // 2, kh // 2, kh // 2, pad_t, 0))
    cache.d[name] = keep
    return F.conv3d(xin, w, b)[0]


def rms_norm(sd, name, x):
    """RMS_norm (vae2_2.py:47-61): L2-normalise over channels, * sqrt(C) * gamma."""
    g = sd[name + ".gamma"].reshape(-1, 1, 1, 1)
    return F.normalize(x, dim=0) * (x.shape[0] ** 0.5) * g


def residual_block(sd, name, x, cache):
    """ResidualBlock (vae2_2.py:195-239): norm-SiLU-conv3 -> norm-SiLU-conv3, + (1x1x1 shortcut | identity)."""
    h = causal_conv(sd, name + ".shortcut", x, cache) if (name + ".shortcut.weight") in sd else x
    y = causal_conv(sd, name + ".residual.2", F.silu(rms_norm(sd, name + ".residual.0", x)), cache)
    y = causal_conv(sd, name + ".residual.6", F.silu(rms_norm(sd, name + ".residual.3", y)), cache)
    return y + h


def attention_block(sd, name, x):
    """AttentionBlock (vae2_2.py:242-283): per frame, single head of width C over the H*W positions."""
    C, T, H, W = x.shape
    xf = x.permute(1, 0, 2, 3)                                        # [T, C, H, W]
    g = sd[name + ".norm.gamma"].reshape(1, -1, 1, 1)
    xn = F.normalize(xf, dim=1) * (C ** 0.5) * g
    qkv = F.conv2d(xn, sd[name + ".to_qkv.weight"], sd[name + ".to_qkv.bias"])       # [T, 3C, H, W]
    q, k, v = qkv.reshape(T, 3 * C, H * W).transpose(1, 2).chunk(3, dim=-1)           # [T, HW, C]
    a = torch.softmax(q @ k.transpose(1, 2) / (C ** 0.5), dim=-1) @ v
    o = F.conv2d(a.transpose(1, 2).reshape(T, C, H, W), sd[name + ".proj.weight"], sd[name + ".proj.bias"])
    return o.permute(1, 0, 2, 3) + x


def resample(sd, name, x, mode, cache, first_chunk):
    """Resample (vae2_2.py:73-171)."""
    C, T, H, W = x.shape
    if mode == "upsample3d" and not first_chunk:                       # first chunk: 'Rep' -> no temporal upsampling
        tc = name + ".time_conv"
        y = causal_conv(sd, tc, x, cache)                              # [2C, T, H, W]  (cache None on the 2nd chunk)
        if cache.d[tc].shape[1] < 2:                                    # 'Rep' is remembered as an explicit zero frame
            cache.d[tc] = torch.cat([torch.zeros_like(cache.d[tc]), cache.d[tc]], dim=1)
        y = y.reshape(2, C, T, H, W)
        x = torch.stack((y[0], y[1]), dim=2).reshape(C, 2 * T, H, W)
    if mode.startswith("upsample"):
        xf = F.interpolate(x.permute(1, 0, 2, 3), scale_factor=(2.0, 2.0), mode="nearest-exact")
        xf = F.conv2d(xf, sd[name + ".resample.1.weight"], sd[name + ".resample.1.bias"], padding=1)
        return xf.permute(1, 0, 2, 3)
    xf = F.conv2d(F.pad(x.permute(1, 0, 2, 3), (0, 1, 0, 1)), sd[name + ".resample.1.weight"],
                  sd[name + ".resample.1.bias"], stride=2).permute(1, 0, 2, 3)
    if mode == "downsample3d":
        tc = name + ".time_conv"
        prev = cache.d.get(tc)
        if prev is None:
            cache.d[tc] = xf.clone()                                    # first chunk: pass through, remember it
        else:
            cache.d[tc] = xf[:, -1:].clone()
            xin = torch.cat([prev[:, -1:], xf], dim=1).unsqueeze(0)
            xf = F.conv3d(xin, sd[tc + ".weight"], sd[tc + ".bias"], stride=(2, 1, 1))[0]
    return xf


def avg_down(x, out_c, ft, fs):
    """AvgDown3D (vae2_2.py:322-373)."""
    C, T, H, W = x.shape
    x = F.pad(x, (0, 0, 0, 0, (ft - T % ft) % ft, 0))
    T = x.shape[1]
    x = x.view(C, T // ft, ft, H // fs, fs, W // fs, fs).permute(0, 2, 4, 6, 1, 3, 5).reshape(C * ft * fs * fs, T // ft, H // fs, W // fs)
    return x.view(out_c, -1, T // ft, H // fs, W // fs).mean(dim=1)


def dup_up(x, out_c, ft, fs, first_chunk):
    """DupUp3D (vae2_2.py:376-418)."""
    C, T, H, W = x.shape
    rep = out_c * ft * fs * fs // C
    x = x.repeat_interleave(rep, dim=0).view(out_c, ft, fs, fs, T, H, W).permute(0, 4, 1, 5, 2, 6, 3)
    x = x.reshape(out_c, T * ft, H * fs, W * fs)
    return x[:, ft - 1:] if first_chunk else x


# ------------------------------------------------------------------------------------------- encoder / decoder passes
def _dims(cfg, enc):
    m = cfg["dim_mult"]
    if enc:
        return [cfg["dim"] * u for u in [1] + m]
    return [cfg["dec_dim"] * u for u in [m[-1]] + m[::-1]]


def encoder_pass(sd, cfg, x, cache, first_chunk):
    """Encoder3d.forward on one chunk (vae2_2.py:565-622 / vae.py:317-366). x [Cin, T, H, W]."""
    v22 = cfg["version"] == "2.2"
    dims, nres, tds = _dims(cfg, True), cfg["num_res_blocks"], cfg["temperal_downsample"]
    x = causal_conv(sd, "encoder.conv1", x, cache)
    li = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        down = i != len(cfg["dim_mult"]) - 1
        t_down = tds[i] if i < len(tds) else False
        if v22:
            x0 = x
            base = f"encoder.downsamples.{i}.downsamples"
            for j in range(nres):
                x = residual_block(sd, f"{base}.{j}", x, cache)
            if down:
                x = resample(sd, f"{base}.{nres}", x, "downsample3d" if t_down else "downsample2d", cache, first_chunk)
            x = x + avg_down(x0, cout, 2 if t_down else 1, 2 if down else 1)
        else:
            for j in range(nres):
                x = residual_block(sd, f"encoder.downsamples.{li}", x, cache)
                li += 1
            if down:
                x = resample(sd, f"encoder.downsamples.{li}", x, "downsample3d" if t_down else "downsample2d", cache, first_chunk)
                li += 1
    x = residual_block(sd, "encoder.middle.0", x, cache)
    x = attention_block(sd, "encoder.middle.1", x)
    x = residual_block(sd, "encoder.middle.2", x, cache)
    return causal_conv(sd, "encoder.head.2", F.silu(rms_norm(sd, "encoder.head.0", x)), cache)


def decoder_pass(sd, cfg, x, cache, first_chunk):
    """Decoder3d.forward on one latent frame (vae2_2.py:681-737 / vae.py:428-472)."""
    v22 = cfg["version"] == "2.2"
    dims, nres = _dims(cfg, False), cfg["num_res_blocks"]
    tus = cfg["temperal_downsample"][::-1]
    x = causal_conv(sd, "decoder.conv1", x, cache)
    x = residual_block(sd, "decoder.middle.0", x, cache)
    x = attention_block(sd, "decoder.middle.1", x)
    x = residual_block(sd, "decoder.middle.2", x, cache)
    li = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        up = i != len(cfg["dim_mult"]) - 1
        t_up = tus[i] if i < len(tus) else False
        mode = "upsample3d" if t_up else "upsample2d"
        if v22:
            x0 = x
            base = f"decoder.upsamples.{i}.upsamples"
            for j in range(nres + 1):
                x = residual_block(sd, f"{base}.{j}", x, cache)
            if up:
                x = resample(sd, f"{base}.{nres + 1}", x, mode, cache, first_chunk)
                x = x + dup_up(x0, cout, 2 if t_up else 1, 2, first_chunk)
        else:
            for j in range(nres + 1):
                x = residual_block(sd, f"decoder.upsamples.{li}", x, cache)
                li += 1
            if up:
                x = resample(sd, f"decoder.upsamples.{li}", x, mode, cache, first_chunk)
                li += 1
    return causal_conv(sd, "decoder.head.2", F.silu(rms_norm(sd, "decoder.head.0", x)), cache)


# ------------------------------------------------------------------------------------------- public
def patchify(x, p):
    """vae2_2.py:286-302  'c f (h q) (w r) -> (c r q) f h w'."""
    if p == 1:
        return x
    C, T, H, W = x.shape
    return x.view(C, T, H // p, p, W // p, p).permute(0, 5, 3, 1, 2, 4).reshape(C * p * p, T, H // p, W // p)


def unpatchify(x, p):
    """vae2_2.py:305-319  '(c r q) f h w -> c f (h q) (w r)'."""
    if p == 1:
        return x
    CC, T, H, W = x.shape
    C = CC // (p * p)
    return x.view(C, p, p, T, H, W).permute(0, 3, 4, 2, 5, 1).reshape(C, T, H * p, W * p)


@torch.no_grad()
def encode(sd, cfg, video):
    """WanVAE_.encode + wrapper (vae2_2.py:797-829,1045-1057): video [3, T, H, W] in [-1,1] -> latent fp32 [z, T', h, w]."""
    mean, inv_std = (t.to(video.device) for t in latent_scale(cfg["version"]))
    x = patchify(video.float(), cfg["patch"])
    T = x.shape[1]
    cache, outs = Cache(), []
    for i in range(1 + (T - 1) // 4):
        chunk = x[:, :1] if i == 0 else x[:, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(encoder_pass(sd, cfg, chunk, cache, i == 0))
    out = torch.cat(outs, dim=1)
    mu = F.conv3d(out.unsqueeze(0), sd["conv1.weight"], sd["conv1.bias"])[0][:cfg["z_dim"]]
    return (mu - mean.view(-1, 1, 1, 1)) * inv_std.view(-1, 1, 1, 1)


@torch.no_grad()
def decode(sd, cfg, z):
    """WanVAE_.decode + wrapper (vae2_2.py:831-860,1059-1072): latent [z, T, h, w] -> video fp32 [3, 1+4(T-1), H, W] in [-1,1]."""
    mean, inv_std = (t.to(z.device) for t in latent_scale(cfg["version"]))
    z = z.float() / inv_std.view(-1, 1, 1, 1) + mean.view(-1, 1, 1, 1)
    x = F.conv3d(z.unsqueeze(0), sd["conv2.weight"], sd["conv2.bias"])[0]
    cache, outs = Cache(), []
    for i in range(x.shape[1]):
        outs.append(decoder_pass(sd, cfg, x[:, i:i + 1], cache, i == 0))
    out = unpatchify(torch.cat(outs, dim=1), cfg["patch"])
    return out.clamp(-1, 1)
