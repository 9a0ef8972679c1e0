"""TEST INFRASTRUCTURE — the pinned oracle (oracle/dit.py, oracle/vae.py) EXECUTED ON THE GPU in fp32: the "device gold" of VERDICT r4 row N2.

Why. The CPU oracle is the specification, but at the sizes bench.py quotes for BASELINE configs[2] (14B, L = 27 810, 40 blocks, CFG: 2.6
PFLOP per step) and for a whole Wan2.2 chunk decode (485 TFLOP) or a production-width Wan2.1 decode (219 TFLOP) it needs hours of host
time. The oracle files are plain torch: the same functions run on `cuda` tensors. This module supplies what that needs and nothing else:

  * `conv_taps`      F.conv3d / F.conv2d as a sum over the kernel taps of [positions, Cin] x [Cin, Cout] fp32 matmuls (rocBLAS sgemm). The
                     image has no MIOpen kernel database for gfx950 (/opt/rocm/share/miopen/db has none): torch's own convolution would
                     compile and search solvers at run time, or fall to a naive kernel. Same sums, another order: ~1e-6 relative.
  * `attention_dev`  attention.py:56-130 semantics (softmax(q k^T / sqrt(D)) v over all keys) one head and one block of queries at a
                     time: fp32 matmul, fp32 softmax — torch's fused SDPA kernels are not used (their fp32 behaviour on ROCm is not ours
                     to vouch for).
  * `on_device(dev)` context manager: oracle.dit / oracle.vae resolve `F.conv3d`, `F.conv2d` and `attention` through the above while it
                     is active; TF32-class shortcuts are switched off.

It is a checker of the checker's own speed problem, not a second specification: every use proves it first against the CPU oracle where
the CPU oracle is affordable (tests/test_zz_full_step_gpu.py: the whole 5B step at L = 9460, asserted <= 1e-4 and printed;
tests/test_zy_vae_fullsize_gpu.py: full-resolution first-latent decodes of both VAEs and a 5-frame encode), and is then used at the sizes
the CPU cannot reach. Only tests/, tools/ and bench.py's parity leg may import it; nothing under yume_amd/ does.
"""
import contextlib
import itertools
import math

import torch
import torch.nn.functional as F


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def conv_taps(x, w, b=None, stride=1, padding=0):
    """F.conv2d / F.conv3d (no dilation, no groups): x [N, Cin, *sp], w [Cout, Cin, *k] -> [N, Cout, *out]. fp32 throughout."""
    nd = w.dim() - 2
    stride, padding = _tup(stride, nd), _tup(padding, nd)
    if any(padding):
        pads = []
        for p in reversed(padding):
            pads += [p, p]
        x = F.pad(x, pads)
    N, Cin = x.shape[:2]
    Cout, ks = w.shape[0], tuple(w.shape[2:])
    osp = tuple((x.shape[2 + i] - ks[i]) // stride[i] + 1 for i in range(nd))
    xc = x.movedim(1, -1).contiguous()                         # [N, *sp, Cin]
    rows = N * math.prod(osp)
    out = None
    for tap in itertools.product(*[range(k) for k in ks]):
        sl = tuple(slice(t, t + (o - 1) * s + 1, s) for t, o, s in zip(tap, osp, stride))
        a = xc[(slice(None),) + sl].reshape(rows, Cin)         # (a copy unless only the leading dims are cut)
        wt = w[(slice(None), slice(None)) + tap].t()           # [Cin, Cout]
        if out is None:
            out = a @ wt
        else:
            out.addmm_(a, wt)
    if b is not None:
        out += b
    return out.view(N, *osp, Cout).movedim(-1, 1).contiguous()      # (the oracle files .view() their activations)


def attention_dev(q, k, v, q_block=8192):
    """q [Lq, N, D], k / v [Lk, N, D] fp32 on one device -> [Lq, N, D]; exact softmax in fp32, one head x q_block queries at a time
    (27 810 keys x 8192 queries = 0.9 GB of scores)."""
    Lq, N, D = q.shape
    out = torch.empty_like(q)
    scale = 1.0 / math.sqrt(D)
    for h in range(N):
        kh, vh = k[:, h].t().contiguous(), v[:, h].contiguous()
        for q0 in range(0, Lq, q_block):
            s = (q[q0:q0 + q_block, h] @ kh).mul_(scale)
            out[q0:q0 + q_block, h] = torch.softmax(s, dim=-1) @ vh
    return out


class _FProxy:
    """torch.nn.functional with the two convolutions the oracle files call rerouted to conv_taps for non-CPU tensors (`force`: for CPU
    tensors too — the plumbing test of this file on the build container)."""

    def __init__(self, force=False):
        self.force = force

    def __getattr__(self, name):
        return getattr(F, name)

    def conv3d(self, x, w, b=None, stride=1, padding=0):
        if x.device.type == "cpu" and not self.force:
            return F.conv3d(x, w, b, stride=stride, padding=padding)
        return conv_taps(x, w, b, stride, padding)

    def conv2d(self, x, w, b=None, stride=1, padding=0):
        if x.device.type == "cpu" and not self.force:
            return F.conv2d(x, w, b, stride=stride, padding=padding)
        return conv_taps(x, w, b, stride, padding)


@contextlib.contextmanager
def on_device(dev="cuda", force=False):
    """while active: oracle.dit / oracle.vae run their convolutions through conv_taps and their attention through attention_dev for
    tensors on `dev`; fp32 matmuls stay fp32 (no TF32-class modes)."""
    from . import dit as odit
    from . import vae as ovae
    saved = (odit.F, ovae.F, odit.attention, torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32,
             torch.get_float32_matmul_precision())
    proxy = _FProxy(force)
    odit.F, ovae.F = proxy, proxy
    cpu_attention = odit.attention
    odit.attention = lambda q, k, v: cpu_attention(q, k, v) if (q.device.type == "cpu" and not force) else attention_dev(q, k, v)
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    torch.set_float32_matmul_precision("highest")
    try:
        yield
    finally:
        odit.F, ovae.F, odit.attention = saved[0], saved[1], saved[2]
        torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32 = saved[3], saved[4]
        torch.set_float32_matmul_precision(saved[5])


def rel_l2(a, b):
    d = a.double() - b.double()
    return (d.norm() / b.double().norm()).item()


# ------------------------------------------------------------------------------------------------ VAE at production size
@torch.no_grad()
def vae_decode(version, z, seed, dev="cuda", cfg=None, force=False):
    """oracle.vae.decode of latent z [zdim, T, h, w] with synth's VAE weights of `seed`, on `dev` in fp32 -> CPU fp32 video."""
    from yume_amd import synth
    from . import vae as ovae
    cfg = cfg or (synth.VAE_CFG_22 if version == "2.2" else synth.VAE_CFG_21)
    sd = synth.make_vae_state_dict(cfg, seed=seed, device=dev)
    with on_device(dev, force):
        out = ovae.decode(sd, cfg, z.to(dev))
    return out.cpu()


@torch.no_grad()
def vae_encode(version, video, seed, dev="cuda", cfg=None, force=False):
    from yume_amd import synth
    from . import vae as ovae
    cfg = cfg or (synth.VAE_CFG_22 if version == "2.2" else synth.VAE_CFG_21)
    sd = synth.make_vae_state_dict(cfg, seed=seed, device=dev)
    with on_device(dev, force):
        out = ovae.encode(sd, cfg, video.to(dev))
    return out.cpu()
