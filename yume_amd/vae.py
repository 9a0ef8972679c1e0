"""Causal 3D VAE engine (Wan2.2 and Wan2.1) on the HIP kernels: chunked encode / decode with the reference's
feat_cache semantics, on channels-last bf16 activations.

Reference behaviour reproduced (wan23/modules/vae2_2.py, wan/modules/vae.py):
  * decode walks the latent one frame at a time (vae2_2.py:839-857), encode in chunks of 1,4,4,... frames (:802-820) — a memory
    measure of the reference: a causal convolution fed [cache ++ chunk] computes the same outputs whatever the chunk length. Here the
    first chunk runs alone (it is the one with different arithmetic: no time_conv in the resamplers) and the following chunks run
    `group` at a time (default 8: 288 GB of HBM hold a 28-frame 704x1280 decode, ~20 GB): at one latent frame per pass the
    44x80 and 88x160 stages launch 56-tile GEMMs on a 256-CU chip, 13 % of the decode time for 3 % of its FLOPs;
  * every temporal conv keeps the last two input frames of the previous chunk (a [2,H,W,C] ring here, passed to the
    conv kernel as its `cache` operand instead of torch.cat + F.pad);
  * upsample3d's time_conv is skipped on the first chunk and starts from a zero cache ('Rep', :116-149);
    downsample3d's strided time_conv is skipped on the first chunk (:159-170);
  * Wan2.2 adds the parameter-free AvgDown3D / DupUp3D shortcuts (:322-418) and patchify(2) (:286-319).
Numerics: bf16 activations and weights, fp32 accumulation, fp32 RMS_norm statistics and softmax. The reference
runs this module in fp32; the stated tolerance is in tests/test_vae_gpu.py and DESIGN.md.
"""
import math
import os

import torch
import torch.nn as nn

from . import ops
from . import vae_ops as V


def _ru(a, b):
    return (a + b - 1) // b * b


def decode_passes(T, group):
    """[(first_latent, end_latent, first_chunk)] for a T-frame latent: the reference walks range(T) one frame per decoder call with
    first_chunk = (i == 0) (vae2_2.py:839-857); here frame 0 alone, then `group` frames per call."""
    return [(0, 1, True)] + [(i, min(T, i + group), False) for i in range(1, T, group)]


def encode_passes(T, group):
    """[(first_frame, end_frame, first_chunk)] for a T-frame video: the reference's chunks are frame 0, then 4 frames each for
    i = 1 .. (T-1)//4 (vae2_2.py:802-820; a ragged tail of < 4 frames is never encoded); here `group` of those chunks per call."""
    nchunk = 1 + (T - 1) // 4
    return [(0, 1, True)] + [(1 + 4 * (i - 1), 1 + 4 * (min(nchunk, i + group) - 1), False) for i in range(1, nchunk, group)]


class _Holder(nn.Module):
    """anonymous container; the tree of these reproduces the reference state_dict key names."""


def build_param_tree(shapes):
    root = _Holder()
    for key, shape in shapes.items():
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _Holder())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape)))
    return root


class VaeEngine:
    def __init__(self, module, cfg):
        self.module = module
        self.cfg = cfg
        self._key = None
        self.P = None
        # reference chunks (latent frames in decode, 4-frame groups in encode) per pass after the first chunk; 1 = the reference's walk
        self.group = max(1, int(os.environ.get("YUME_VAE_GROUP", "8")))

    # ------------------------------------------------------------------ weights
    def _pack(self):
        sd = {k: v.detach() for k, v in self.module.state_dict().items()}
        dev = next(iter(sd.values())).device
        if dev.type != "cuda":
            raise RuntimeError("yume_amd VAE must live on the GPU ('cuda') — there is no CPU path")
        P = {}
        for k, w in sd.items():
            if k.endswith(".weight"):
                name = k[:-7]
                if w.dim() == 4:            # Conv2d -> kt = 1
                    w = w.unsqueeze(2)
                co, ci, kt, kh, kw = w.shape
                cip = _ru(ci, 8)
                cop = _ru(co, 4)
                wt = w.float().permute(0, 2, 3, 4, 1)                       # [co, kt, kh, kw, ci]
                if cip != ci:
                    wt = torch.cat([wt, wt.new_zeros(co, kt, kh, kw, cip - ci)], dim=-1)
                wt = wt.reshape(co, kt * kh * kw * cip)
                K = wt.shape[1]
                Kp = _ru(K, 64)
                full = torch.zeros(cop, Kp, dtype=torch.float32, device=dev)
                full[:co, :K] = wt
                b = torch.zeros(cop, dtype=torch.float32, device=dev)
                b[:co] = sd[name + ".bias"].float()
                P[name] = dict(w=full.to(torch.bfloat16).contiguous(), b=b, co=co, cop=cop, ci=ci, cip=cip, k=(kt, kh, kw))
            elif k.endswith(".gamma"):
                P[k] = w.float().reshape(-1).contiguous()
        self.P = P
        self.dev = dev
        self.zero = torch.zeros(64, dtype=torch.bfloat16, device=dev)

    def ensure_packed(self):
        key = tuple((p.data_ptr(), p._version) for p in self.module.parameters())
        if key != self._key:
            self._pack()
            self._key = key

    # ------------------------------------------------------------------ primitives
    def _new(self, T, H, W, C):
        return torch.empty((T, H, W, C), dtype=torch.bfloat16, device=self.dev)

    def _update_cache(self, cache, name, x):
        """cache[name] <- last two frames of (previous cache ++ x); created zero-filled on first use."""
        if x.shape[0] >= 2:
            # no copy: the last two frames of this chunk's input ARE the next chunk's cache (x is never written again;
            # holding the view keeps it alive — 288 GB of HBM make that free)
            cache[name] = x[-2:]
            return
        c = cache.get(name)
        if c is None:
            c = torch.zeros((2,) + tuple(x.shape[1:]), dtype=torch.bfloat16, device=self.dev)
        else:
            c = torch.stack((c[1], x[0]))      # [previous last frame, this frame] in fresh memory (c may be a view)
            cache[name] = c
            return
        c[1].copy_(x[0])
        cache[name] = c

    def _conv3(self, name, x, cache, add=None, norm=None):
        """CausalConv3d with a temporal kernel (3x3x3 or (3,1,1)), stride 1: reads cache[name], then updates it.
        norm: the name of an RMS_norm (+ SiLU) applied to the convolution's output inside the same C-ABI call (YUME_CONV_EPI_RMS_SILU)."""
        p = self.P[name]
        kt, kh, kw = p["k"]
        T, H, W, C = x.shape
        out = self._new(T, H, W, _ru(p["co"], 8))
        if out.shape[3] != p["cop"]:
            out.zero_()
        if norm is not None:
            if add is not None or out.shape[3] != p["cop"] or p["cop"] != p["co"]:
                raise RuntimeError("yume_amd.vae: the fused norm needs an unpadded channel row and no shortcut")
            epi, operand = V.EPI_RMS_SILU, self.P[norm + ".gamma"]
        else:
            epi, operand = (V.EPI_ADD if add is not None else V.EPI_BF16), add
        V.conv3d_cl(x, cache.get(name), p["w"], p["b"], p["cop"], (kt, kh, kw), (1, 1, 1), (kt - 1, kh // 2, kw // 2), False,
                    out, epi, add=operand, zero_page=self.zero)
        self._update_cache(cache, name, x)
        return out

    def _conv1(self, name, x, add=None):
        """1x1x1 CausalConv3d / 1x1 Conv2d."""
        p = self.P[name]
        T, H, W, C = x.shape
        out = self._new(T, H, W, _ru(p["co"], 8))
        if out.shape[3] != p["cop"]:
            out.zero_()
        V.conv3d_cl(x, None, p["w"], p["b"], p["cop"], (1, 1, 1), (1, 1, 1), (0, 0, 0), False, out,
                    V.EPI_ADD if add is not None else V.EPI_BF16, add=add, zero_page=self.zero)
        return out

    def _norm(self, name, x, silu=True):
        out = torch.empty_like(x)
        return V.rmsnorm_silu(x, self.P[name + ".gamma"], silu, out)

    def _res(self, name, x, cache):
        h = self._conv1(name + ".shortcut", x) if (name + ".shortcut") in self.P else x
        # residual.3 (RMS_norm) + residual.4 (SiLU) ride in the epilogue of residual.2's call: conv -> norm -> SiLU is one C-ABI call
        # (fused in the kernel at the 96 / 160-channel levels, conv + in-place norm kernel elsewhere; include/yume_hip.h)
        yn = self._conv3(name + ".residual.2", self._norm(name + ".residual.0", x), cache, norm=name + ".residual.3")
        return self._conv3(name + ".residual.6", yn, cache, add=h)

    def _attn(self, name, x):
        """AttentionBlock: per frame, one head of width C over the H*W positions (two GEMMs + a row softmax)."""
        T, H, W, C = x.shape
        hw = H * W
        if C % 64:
            raise RuntimeError("yume_amd VAE attention needs C % 64 == 0")
        pq = self.P[name + ".to_qkv"]
        xn = self._norm(name + ".norm", x, silu=False)
        out = torch.empty_like(x)
        hwp, hw4 = _ru(hw, 64), _ru(hw, 4)     # K padding of the P.V product; N padding of the score GEMM (zero key rows)
        qk = torch.zeros((hw4, 2 * C), dtype=torch.bfloat16, device=self.dev)
        vt = torch.zeros((C, hwp), dtype=torch.bfloat16, device=self.dev)
        s = torch.empty((hw, hw4), dtype=torch.float32, device=self.dev)
        pm = torch.empty((hw, hwp), dtype=torch.bfloat16, device=self.dev)
        o = self._new(1, H, W, C)
        for t in range(T):
            ops.gemm_bf16(xn[t].view(hw, C), pq["w"], pq["b"], qk[:hw], ops.EPI_BF16_SPLITT, out_t=vt, n_split=2 * C)
            ops.gemm_bf16(qk[:hw, :C], qk[:, C:], None, s, ops.EPI_F32)
            V.softmax_rows(s, hw, 1.0 / math.sqrt(C), pm)
            ops.gemm_bf16(pm, vt, None, o.view(hw, C), ops.EPI_BF16)
            pp = self.P[name + ".proj"]
            V.conv3d_cl(o, None, pp["w"], pp["b"], pp["cop"], (1, 1, 1), (1, 1, 1), (0, 0, 0), False, out[t:t + 1],
                        V.EPI_ADD, add=x[t:t + 1], zero_page=self.zero)
        return out

    def _upsample(self, name, x, mode, cache, first_chunk):
        T, H, W, C = x.shape
        if mode == "upsample3d" and not first_chunk:
            tc = name + ".time_conv"
            p = self.P[tc]
            y = self._new(2 * T, H, W, C)
            V.conv3d_cl(x, cache.get(tc), p["w"], p["b"], p["cop"], (3, 1, 1), (1, 1, 1), (2, 0, 0), False, y, V.EPI_TSPLIT,
                        zero_page=self.zero)
            self._update_cache(cache, tc, x)
            x, T = y, 2 * T
        p = self.P[name + ".resample.1"]
        out = self._new(T, 2 * H, 2 * W, _ru(p["co"], 8))
        V.conv3d_cl(x, None, p["w"], p["b"], p["cop"], (1, 3, 3), (1, 1, 1), (0, 1, 1), True, out, V.EPI_BF16, zero_page=self.zero)
        return out

    def _downsample(self, name, x, mode, cache, first_chunk):
        T, H, W, C = x.shape
        p = self.P[name + ".resample.1"]
        Ho, Wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
        y = self._new(T, Ho, Wo, C)
        V.conv3d_cl(x, None, p["w"], p["b"], p["cop"], (1, 3, 3), (1, 2, 2), (0, 0, 0), False, y, V.EPI_BF16, zero_page=self.zero)
        if mode == "downsample3d":
            tc = name + ".time_conv"
            if first_chunk:
                self._update_cache(cache, tc, y)          # pass through; remember the frame
                return y
            p = self.P[tc]
            To = (T + 1 - 3) // 2 + 1
            z = self._new(To, Ho, Wo, C)
            V.conv3d_cl(y, cache.get(tc), p["w"], p["b"], p["cop"], (3, 1, 1), (2, 1, 1), (1, 0, 0), False, z, V.EPI_BF16,
                        zero_page=self.zero)
            self._update_cache(cache, tc, y)
            return z
        return y

    def _dims(self, enc):
        m = self.cfg["dim_mult"]
        return [self.cfg["dim"] * u for u in [1] + m] if enc else [self.cfg["dec_dim"] * u for u in [m[-1]] + m[::-1]]

    # ------------------------------------------------------------------ passes
    def _encoder_pass(self, x, cache, first_chunk):
        cfg = self.cfg
        v22 = cfg["version"] == "2.2"
        dims, nres, tds = self._dims(True), cfg["num_res_blocks"], cfg["temperal_downsample"]
        x = self._conv3("encoder.conv1", x, cache)
        li = 0
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            down = i != len(cfg["dim_mult"]) - 1
            t_down = tds[i] if i < len(tds) else False
            mode = "downsample3d" if t_down else "downsample2d"
            if v22:
                x0, base = x, f"encoder.downsamples.{i}.downsamples"
                for j in range(nres):
                    x = self._res(f"{base}.{j}", x, cache)
                if down:
                    x = self._downsample(f"{base}.{nres}", x, mode, cache, first_chunk)
                V.avgdown_add(x0, x, 2 if t_down else 1, 2 if down else 1)
            else:
                for j in range(nres):
                    x = self._res(f"encoder.downsamples.{li}", x, cache)
                    li += 1
                if down:
                    x = self._downsample(f"encoder.downsamples.{li}", x, mode, cache, first_chunk)
                    li += 1
        x = self._res("encoder.middle.0", x, cache)
        x = self._attn("encoder.middle.1", x)
        x = self._res("encoder.middle.2", x, cache)
        return self._conv3("encoder.head.2", self._norm("encoder.head.0", x), cache)

    def _decoder_pass(self, x, cache, first_chunk):
        cfg = self.cfg
        v22 = cfg["version"] == "2.2"
        dims, nres = self._dims(False), cfg["num_res_blocks"]
        tus = cfg["temperal_downsample"][::-1]
        x = self._conv3("decoder.conv1", x, cache)
        x = self._res("decoder.middle.0", x, cache)
        x = self._attn("decoder.middle.1", x)
        x = self._res("decoder.middle.2", x, cache)
        li = 0
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            up = i != len(cfg["dim_mult"]) - 1
            t_up = tus[i] if i < len(tus) else False
            mode = "upsample3d" if t_up else "upsample2d"
            if v22:
                x0, base = x, f"decoder.upsamples.{i}.upsamples"
                for j in range(nres + 1):
                    x = self._res(f"{base}.{j}", x, cache)
                if up:
                    x = self._upsample(f"{base}.{nres + 1}", x, mode, cache, first_chunk)
                    ft = 2 if t_up else 1
                    V.dupup_add(x0, x, ft, 2, (ft - 1) if first_chunk else 0)
            else:
                for j in range(nres + 1):
                    x = self._res(f"decoder.upsamples.{li}", x, cache)
                    li += 1
                if up:
                    x = self._upsample(f"decoder.upsamples.{li}", x, mode, cache, first_chunk)
                    li += 1
        return self._conv3("decoder.head.2", self._norm("decoder.head.0", x), cache)

    # ------------------------------------------------------------------ public
    @torch.no_grad()
    def _vec(self, v, n):
        if v is None:
            return None
        if not isinstance(v, torch.Tensor):
            v = torch.full((n,), float(v))
        return v.to(device=self.dev, dtype=torch.float32).reshape(-1).expand(n).contiguous()

    def encode(self, video, sub=None, mul=None):
        """video [3, T, H, W] (fp32|bf16, [-1,1]) -> latent fp32 [z, 1+(T-1)//4, H/s, W/s] = (mu - sub) * mul —
        WanVAE_.encode (sub = mean, mul = 1/std)."""
        self.ensure_packed()
        cfg = self.cfg
        ps, z = cfg["patch"], cfg["z_dim"]
        v = video.to(self.dev)
        if v.dtype not in (torch.float32, torch.bfloat16):
            v = v.float()
        v = v.contiguous()
        C, T, H, W = v.shape
        cpad = _ru(C * ps * ps, 8)
        x = self._new(T, H // ps, W // ps, cpad)
        V.pack_input(v, ps, None, None, x)
        cache, outs = {}, []
        for a, b, first in encode_passes(T, self.group):
            outs.append(self._encoder_pass(x[a:b], cache, first))
        out = torch.cat(outs, dim=0)                                     # [T', h, w, 2z(pad)]
        mu = self._conv1("conv1", out)
        Tl, h, w, _ = mu.shape
        res = torch.empty((z, Tl, h, w), dtype=torch.float32, device=self.dev)
        if (sub is None) != (mul is None):
            sub, mul = (0.0 if sub is None else sub), (1.0 if mul is None else mul)
        V.unpack_output(mu, z, 1, self._vec(sub, z), self._vec(mul, z), 0.0, 0.0, res)
        return res

    @torch.no_grad()
    def decode(self, zlat, mul=None, add=None, clamp=True):
        """latent [z, T, h, w] -> video fp32 [3, 1+4(T-1), H, W] — WanVAE_.decode on z*mul + add (mul = std,
        add = mean), clamped to [-1,1] like the Wan*VAE wrappers when clamp."""
        self.ensure_packed()
        cfg = self.cfg
        ps, z = cfg["patch"], cfg["z_dim"]
        zl = zlat.to(self.dev)
        if zl.dtype not in (torch.float32, torch.bfloat16):
            zl = zl.float()
        zl = zl.contiguous()
        C, T, h, w = zl.shape
        if C != z:
            raise RuntimeError(f"latent has {C} channels, VAE z_dim is {z}")
        x = self._new(T, h, w, _ru(z, 8))
        if (mul is None) != (add is None):
            mul, add = (1.0 if mul is None else mul), (0.0 if add is None else add)
        V.pack_input(zl, 1, self._vec(mul, z), self._vec(add, z), x)
        x = self._conv1("conv2", x)
        cache, outs = {}, []
        for a, b, first in decode_passes(T, self.group):
            outs.append(self._decoder_pass(x[a:b], cache, first))
        out = torch.cat(outs, dim=0)                                     # [T_out, H/ps, W/ps, in_ch(pad)]
        To, Ho, Wo, _ = out.shape
        cv = cfg["in_ch"]
        res = torch.empty((cv // (ps * ps), To, Ho * ps, Wo * ps), dtype=torch.float32, device=self.dev)
        V.unpack_output(out, cv, ps, None, None, -1.0 if clamp else 0.0, 1.0 if clamp else 0.0, res)
        return res
