"""TEST INFRASTRUCTURE — runs the denoise loops of the reference's sampling SCRIPTS themselves (not a restatement).

The loops live inside `main()`-style functions of scripts that cannot be imported (argparse, datasets, checkpoints and
`torch.distributed` set-up at import / call time):

    fastvideo/sample/sample_5b.py   :958-1034   Euler ODE, 5B, clean history, per-token timesteps built from mask2
    fastvideo/sample/sample.py      :767-790    Euler ODE, 14B, CFG 5.0, history re-noised with the next sigma
    fastvideo/sample/sample_tts.py  :690-868    SDE (eta 0.3) + time travel (step 2, interval 2), 14B, CFG 5.0

So the loop is cut out of the script's TEXT (by its own anchor lines, never by copying it into this repository), dedented and
executed in a namespace that supplies exactly the names the loop reads: a stand-in `transformer`, the tensors, the sigma schedule and
a `torch` whose `randn_like` draws are recorded so that a restatement can replay them. That pins `oracle/sampler.py` (and through it
`yume_amd/sampling.py`) to the scripts' real control flow, including their quirks (the stale `current_pred`, the `i + 1 == 50` test).
Only usable where /root/reference exists (the build container); `oracle/make_golden_sampler.py` stores what it produces as a fixture.
"""
import math
import os
import textwrap

import torch

REF = os.environ.get("YUME_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "fastvideo", "sample", "sample_tts.py"))


def _cut(path, first_anchor, loop_head="for i in range(sample_step):", include_from_anchor=False):
    """the `for i in range(sample_step):` loop behind the first line containing `first_anchor` (optionally starting AT that line, for the
    configuration assignments in front of the loop), up to the first line that is not indented deeper than the loop head."""
    src = open(path).read().split("\n")
    a = next(i for i, l in enumerate(src) if first_anchor in l)
    h = next(i for i in range(a, len(src)) if src[i].strip() == loop_head)
    ind = len(src[h]) - len(src[h].lstrip())
    e = h + 1
    while e < len(src) and (src[e].strip() == "" or len(src[e]) - len(src[e].lstrip()) > ind):
        e += 1
    start = a if include_from_anchor else h
    lines = [l for l in src[start:e]]
    # the lines between the anchor and the loop head must sit at the loop's indentation (plain assignments)
    assert all(l.strip() == "" or len(l) - len(l.lstrip()) >= ind for l in lines)
    return textwrap.dedent("\n".join(lines)), (start + 1, e)


class _Torch:
    """`torch` for the executed loop: everything is torch's, except that randn_like draws are recorded (float64, seeded generator)."""

    def __init__(self, seed):
        self._g = torch.Generator().manual_seed(seed)
        self.draws = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def randn_like(self, x):
        r = torch.randn(x.shape, generator=self._g, dtype=x.dtype)
        self.draws.append(r)
        return r


def _sigma_index(t, sigmas):
    """which sigma the loop asked for: the index whose 1000*sigma is nearest to the (last entry of the) timestep tensor"""
    v = float(t.flatten()[-1])
    return min(range(len(sigmas)), key=lambda i: abs(sigmas[i] * 1000.0 - v))


def run_tts(f, latent, noise, model_input, sigmas, lfz, sde=True, step_sample=0, seed=7):
    """sample_tts.py's loop with `transformer(...)[0] = f(latent, sigma_index, "cond" | "uncond")`.
    Returns (final latent, list of randn_like draws in order, list of (sigma_index, which) calls, (first, last) line numbers)."""
    code, span = _cut(os.path.join(REF, "fastvideo", "sample", "sample_tts.py"), "time_travel_step = 2", include_from_anchor=True)
    calls = []

    def transformer(latent_model_input, t=None, rand_num_img=None, latent_frame_zero=None, which=None):
        assert rand_num_img == 0.6 and latent_frame_zero == lfz and len(latent_model_input) == 1
        i = _sigma_index(t, sigmas)
        calls.append((i, which))
        return (f(latent_model_input[0], i, which), None)

    tp = _Torch(seed)
    ns = dict(torch=tp, math=math, sample_step=len(sigmas), sampling_sigmas=sigmas, latent=latent, noise=noise, model_input=model_input,
              model_input_1=model_input, step_sample=step_sample, sde=sde, device="cpu", transformer=transformer, rand_num_img=0.6,
              latent_frame_zero=lfz, arg_c={"which": "cond"}, arg_null={"which": "uncond"})
    exec(compile(code, "sample_tts.py[%d:%d]" % span, "exec"), ns)
    return ns["latent"], tp.draws, calls, span


def run_euler_14b(f, latent, noise, model_input, sigmas, lfz, step_sample=0):
    """sample.py's loop (CFG 5.0, history re-noised)."""
    code, span = _cut(os.path.join(REF, "fastvideo", "sample", "sample.py"), "sampling_sigmas = get_sampling_sigmas(sample_step, 3.0)")
    calls = []

    def transformer(latent_model_input, t=None, rand_num_img=None, which=None):
        assert rand_num_img == 0.6 and len(latent_model_input) == 1
        i = _sigma_index(t, sigmas)
        calls.append((i, which))
        return f(latent_model_input[0], i, which), None

    ns = dict(torch=torch, math=math, sample_step=len(sigmas), sampling_sigmas=sigmas, latent=latent, noise=noise, model_input=model_input,
              model_input_1=model_input, step_sample=step_sample, device="cpu", transformer=transformer, rand_num_img=0.6,
              latent_frame_zero=lfz, arg_c={"which": "cond"}, arg_null={"which": "uncond"})
    exec(compile(code, "sample.py[%d:%d]" % span, "exec"), ns)
    return ns["latent"], calls, span


def run_euler_5b(f, latent, model_input, sigmas, lfz, seq_len, step_sample=0):
    """sample_5b.py's loop (i2v / later chunks: `not t2v or step_sample > 0`): clean history, per-token timestep vector built from mask2
    exactly as the script does. mask2 is what wan23's masks_like(zero=True) hands the script: zeros on the history frames, ones on the
    frames being denoised (wan23/utils/utils.py:106-133). Returns (final latent, list of timestep tensors, calls, span)."""
    code, span = _cut(os.path.join(REF, "fastvideo", "sample", "sample_5b.py"), "sampling_sigmas = get_sampling_sigmas(sample_step, 7.0)")
    calls, tvecs = [], []
    mask = torch.ones_like(latent)
    mask[:, :-lfz] = 0

    def transformer(latent_model_input, t=None, seq_len=None, which=None, flag=True):
        assert flag is True and len(latent_model_input) == 1
        i = _sigma_index(t, sigmas)
        calls.append((i, which))
        tvecs.append(t.clone())
        return [f(latent_model_input[0], i, which)]

    ns = dict(torch=torch, math=math, sample_step=len(sigmas), sampling_sigmas=sigmas, latent=latent, model_input=model_input,
              model_input_1=model_input, step_sample=step_sample, t2v=False, device="cpu", transformer=transformer, mask2=[mask],
              latent_frame_zero=lfz, arg_c={"which": "cond", "seq_len": seq_len}, print=lambda *a, **k: None)
    exec(compile(code, "sample_5b.py[%d:%d]" % span, "exec"), ns)
    return ns["latent"], tvecs, calls, span


# ---- shared by oracle/make_golden_sampler.py and tests/test_sampling.py (needs no reference tree) -------------------------------------
def script_field(seed, channels):
    """a deterministic stand-in velocity field f(latent [C,F,H,W], sigma_index, "cond" | "uncond") -> [C,F,H,W]"""
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(channels, channels, generator=g, dtype=torch.float64) * 0.3

    def f(latent, i, which):
        s = 1.0 if which == "cond" else 0.7
        return torch.tanh(torch.einsum("cd,dfhw->cfhw", w.to(latent.dtype), latent)) * s + 0.01 * i
    return f


def script_case(seed, dtype, C=6, F=7, H=4, W=6):
    g = torch.Generator().manual_seed(seed)
    mk = lambda: torch.randn(C, F, H, W, generator=g, dtype=torch.float64).to(dtype)
    return dict(model_input=mk(), noise=mk(), C=C, F=F, H=H, W=W)


# ---- the web app's frame conversion (webapp_single_gpu.py:117-121) ----------------------------------------------------------------------
def run_webapp_postprocess(video):
    """executes the reference's `_postprocess_video(video, fps, out_path)` itself (cut out of webapp_single_gpu.py, which cannot be
    imported: gradio, the model stack and CUDA at import) with `export_to_video` replaced by a recorder.
    video: fp32 [C,F,H,W] -> numpy uint8 [F,H,W,C] (the frames it would have encoded)."""
    import numpy as np
    from PIL import Image
    src = open(os.path.join(REF, "webapp_single_gpu.py")).read().split("\n")
    h = next(i for i, l in enumerate(src) if l.startswith("def _postprocess_video("))
    e = h + 1
    while e < len(src) and (src[e].strip() == "" or src[e].startswith((" ", "\t"))):
        e += 1
    got = {}
    ns = dict(torch=torch, np=np, Image=Image, export_to_video=lambda frames, out_path, fps=None: got.update(frames=frames, fps=fps))
    exec(compile("\n".join(src[h:e]), "webapp_single_gpu.py[%d:%d]" % (h + 1, e), "exec"), ns)      # (the decorator line above it is not needed)
    ns["_postprocess_video"](video, 16, "unused.mp4")
    return np.stack([np.asarray(f) for f in got["frames"]])


def webapp_case(seed=3, shape=(3, 5, 16, 24)):
    """seeded video with every k/255 and k/255 +- one ulp boundary of the truncating conversion among its samples, values outside [-1, 1]"""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(shape, generator=g) * 0.8
    flat = v.view(-1)
    k = torch.arange(256, dtype=torch.float64)
    edge = ((k / 255) * 2 - 1).float()
    pts = torch.cat([edge, torch.nextafter(edge, torch.tensor(2.0)), torch.nextafter(edge, torch.tensor(-2.0)), torch.tensor([-1.5, 1.5, -1.0, 1.0, 0.0])])
    flat[:pts.numel()] = pts
    return v
