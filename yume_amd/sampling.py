"""Denoise-loop drivers around WanModel.forward (host side; the per-element updates are negligible next to the
forward and stay torch). They mirror the loops the reference keeps inside its scripts:

  ode_chunk        fastvideo/sample/sample_5b.py:942-1034 (5B, no CFG, clean history) and
                   fastvideo/sample/sample.py:745-790 (14B, CFG 5.0, history re-noised every step)
  sde_tts_chunk    fastvideo/sample/sample_tts.py:694-868 (SDE noise eta 0.3 + "time travel" look-ahead), including its
                   control-flow quirks: the look-ahead prediction of the LAST travelled step is reused
                   when the look-ahead range is empty, and the SDE branch tests `i + 1 == 50` instead of the step count.
  long_video       the FramePack chunk loop of sample_5b.py:920-1097 (history grows by 8 latent frames per chunk, VAE decode
                   of the 8 new latents per chunk).

A `velocity(latent, i)` callable hides the model call (+CFG); `history(index)` returns the history latents the
sampler concatenates in front of the frames being denoised for sigma index `index` (clean for 5B, re-noised for 14B).
"""
import math
from typing import Callable, List, Optional

import torch

from .synth import sampling_sigmas  # noqa: F401  (sample_5b.py:502-506)


def make_velocity_5b(model, context, seq_len, n_hist_tok, n_new_tok, sigmas, lfz=8):
    """5B forward with the per-token timestep vector of sample_5b.py:965-972 (0 on history tokens, 1000*sigma_i on the rest)."""
    dev = next(model.parameters()).device
    zeros = torch.zeros(n_hist_tok, dtype=torch.float64, device=dev)
    ones = torch.ones(n_new_tok, dtype=torch.float64, device=dev)

    def velocity(latent, i):
        t = torch.cat([zeros, ones * (sigmas[i] * 1000.0)]).unsqueeze(0)
        return model([latent], t=t, context=context, seq_len=seq_len, latent_frame_zero=lfz, flag=True)[0]
    return velocity


def make_velocity_14b(model, arg_c, arg_null, sigmas, guide=5.0, rand_num_img=0.6, lfz=None):
    """14B forward with classifier-free guidance: uncond + 5.0 * (cond - uncond) (sample.py:774-779)."""
    dev = next(model.parameters()).device
    kw = {} if lfz is None else {"latent_frame_zero": lfz}

    def velocity(latent, i):
        t = torch.tensor([sigmas[i] * 1000.0], device=dev)
        c = model([latent], t=t, rand_num_img=rand_num_img, **kw, **arg_c)[0]
        u = model([latent], t=t, rand_num_img=rand_num_img, **kw, **arg_null)[0]
        return u + guide * (c - u)
    return velocity


def clean_history(hist):
    """5B: the history latents stay clean at every step (sample_5b.py:1031-1034)."""
    return lambda index: hist


def renoised_history(hist, noise_hist, sigmas):
    """14B: noise*sigma + (1-sigma)*history at the NEXT sigma index (sample.py:786-790)."""
    return lambda index: noise_hist * sigmas[index] + (1 - sigmas[index]) * hist


@torch.no_grad()
def ode_chunk(velocity: Callable, latent, sigmas: List[float], lfz: int, history: Callable):
    """Euler ODE over all sigmas; latent = cat([history, x]) on the frame axis (dim 1). Returns the final latent."""
    S = len(sigmas)
    for i in range(S):
        v = velocity(latent, i)
        nxt = sigmas[i + 1] if i + 1 < S else 0.0
        x = latent[:, -lfz:] + (nxt - sigmas[i]) * v[:, -lfz:]
        latent = torch.cat([history(min(S - 1, i + 1)), x], dim=1)
    return latent


def _sde_step(x, v, s, s_next, eta, last_is_50, gen):
    """the SDE branch of sample_tts.py:726-744 / :806-818 applied to the Euler mean x + (s_next - s) v."""
    mean = x + (s_next - s) * v
    x0 = x + (0 - s) * v
    delta_t = 0.0 if last_is_50 else max(s - s_next, 0.0)
    dsigma = (0 - s) if last_is_50 else (s_next - s)
    score = -(x - x0 * (1 - s)) / s ** 2
    mean = mean + (-0.5 * eta ** 2 * score) * dsigma
    noise = torch.randn(mean.shape, generator=gen, device=mean.device, dtype=mean.dtype)
    return mean + noise * (eta * math.sqrt(delta_t))


@torch.no_grad()
def sde_tts_chunk(velocity: Callable, latent, sigmas: List[float], lfz: int, history: Callable, sde=True, eta=0.3,
                  travel_step=2, travel_interval=2, generator: Optional[torch.Generator] = None, i0=0, i1=None, current_pred=None,
                  return_state=False):
    """sample_tts.py:694-868. `velocity(latent, i)` is evaluated at sigma index i; returns the final latent.
    i0 / i1 / current_pred / return_state run the sampler steps [i0, i1) only and hand the look-ahead state on (bench.py times a
    window of steps of the 50-step loop); the defaults run the whole chunk."""
    S = len(sigmas)
    for i in range(i0, S if i1 is None else min(i1, S)):
        v = velocity(latent, i)
        x = latent[:, -lfz:]
        nxt = sigmas[i + 1] if i + 1 < S else 0.0
        temp = x + (nxt - sigmas[i]) * v[:, -lfz:]
        if sde:
            # NB the reference tests `i + 1 == 50`, not the step count (:730,736): with S != 50 the last step uses
            # sigmas[i+1], which does not exist there; we follow the reference for S == 50 and treat the out-of-range
            # sigma as 0 otherwise (identical value: the schedule ends at 0).
            temp = _sde_step(x, v[:, -lfz:], sigmas[i], nxt, eta, (i + 1 == 50), generator)
        if travel_interval > 0 and i % travel_interval == 0:
            stop = min(S - 1, i + travel_step)
            lt = torch.cat([history(stop), temp], dim=1)
            for j in range(i + 1, stop):
                vt = velocity(lt, j)
                xt = lt[:, -lfz:]
                tt = xt + (sigmas[j + 1] - sigmas[j]) * vt[:, -lfz:]
                if sde:
                    tt = _sde_step(xt, vt[:, -lfz:], sigmas[j], sigmas[j + 1], eta, False, generator)
                lt = torch.cat([history(min(S - 1, j + 1)), tt], dim=1)
                current_pred = vt
            if current_pred is None:
                raise RuntimeError("time travel needs at least one look-ahead step before the first reuse "
                                   "(the reference would hit an unbound `current_pred`)")
            # step i is REDONE with the look-ahead velocity (stale from the last travelled step when the range was empty)
            temp = x + (nxt - sigmas[i]) * current_pred[:, -lfz:]
        latent = torch.cat([history(min(S - 1, i + 1)), temp], dim=1)
    return (latent, current_pred) if return_state else latent


def tts_forward_count(S, travel_step=2, travel_interval=2, i0=0, i1=None):
    """number of velocity evaluations of sde_tts_chunk over the sampler steps [i0, i1) (S=50 -> 74)."""
    i1 = S if i1 is None else min(i1, S)
    n = i1 - i0
    for i in range(i0, i1):
        if travel_interval > 0 and i % travel_interval == 0:
            n += max(0, min(S - 1, i + travel_step) - (i + 1))
    return n


@torch.no_grad()
def long_video_5b(model, vae, first_history, contexts, steps, shift=7.0, lfz=8, generator=None, decode=True,
                  on_chunk: Optional[Callable] = None):
    """FramePack long-video loop (sample_5b.py:920-1097): for each caption one 2-second chunk = `steps` Euler steps on
    [history | 8 noisy latents], then the 8 new latents are appended to the history and VAE-decoded.
    first_history: latents [48, F0, H, W] (the encoded conditioning clip). Returns (all latents, list of decoded chunks)."""
    from . import framepack
    sig = sampling_sigmas(steps, shift)
    hist = first_history
    dev = hist.device
    videos = []
    for k, ctx in enumerate(contexts):
        C, F0, H, W = hist.shape
        plan = framepack.pack_plan(F0 + lfz, H, W, lfz)
        noise = torch.randn((C, lfz, H, W), generator=generator, device=dev, dtype=hist.dtype)
        latent = torch.cat([hist, noise], dim=1)
        vel = make_velocity_5b(model, [ctx], plan.seq_len, plan.n_hist_tok, plan.n_new_tok, sig, lfz)
        latent = ode_chunk(vel, latent, sig, lfz, clean_history(hist))
        hist = latent                                       # the history of the next chunk includes the new frames
        if decode and vae is not None:
            videos.append(vae.decode([latent[:, -lfz:]])[0])
        if on_chunk is not None:
            on_chunk(k, latent)
    return hist, videos
