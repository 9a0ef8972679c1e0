"""TEST INFRASTRUCTURE — CPU restatement of the reference's denoise loops, kept deliberately close to the scripts'
own variable flow so it is an independent check of yume_amd/sampling.py:

  euler_5b    fastvideo/sample/sample_5b.py:942-1034   (clean history, one forward per step)
  euler_14b   fastvideo/sample/sample.py:745-790       (CFG 5.0, history re-noised with sigma_{i+1})
  tts         fastvideo/sample/sample_tts.py:694-868   (SDE eta 0.3, time travel step 2 / interval 2 / repeat 1)

PINNED: the scripts cannot be imported, but their loops can be cut out of the script text and executed — oracle/ref_scripts.py does
that; tests/test_sampling.py holds this restatement BIT-IDENTICAL to the scripts' own loops (live in the build container, and through
tests/golden/sampler_scripts.pt, written by oracle/make_golden_sampler.py, anywhere else), including the order and number of model calls.

`transformer(latent, sigma_index, which)` stands for the model call (which in {"cond","uncond"}) and returns the velocity
for the whole latent; `randn(shape)` stands for torch.randn_like so tests can replay the same noise.
"""
import math

import numpy as np
import torch


def get_sampling_sigmas(sampling_steps, shift):
    """sample_5b.py:502-506 (numpy, float64)."""
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return (shift * sigma / (1 + (shift - 1) * sigma))


def euler_5b(transformer, latent, model_input, sigmas, lfz):
    S = len(sigmas)
    for i in range(S):
        pred = transformer(latent, i, "cond")
        if i + 1 == S:
            temp_x0 = latent[:, -lfz:] + (0 - sigmas[i]) * pred[:, -lfz:]
        else:
            temp_x0 = latent[:, -lfz:] + (sigmas[i + 1] - sigmas[i]) * pred[:, -lfz:]
        latent = torch.cat([model_input[:, :-lfz], temp_x0], dim=1)
    return latent


def euler_14b(transformer, latent, model_input, noise, sigmas, lfz, guide=5.0):
    S = len(sigmas)
    for i in range(S):
        cond = transformer(latent, i, "cond")
        uncond = transformer(latent, i, "uncond")
        pred = uncond + guide * (cond - uncond)
        if i + 1 == S:
            temp_x0 = latent[:, -lfz:] + (0 - sigmas[i]) * pred[:, -lfz:]
        else:
            temp_x0 = latent[:, -lfz:] + (sigmas[i + 1] - sigmas[i]) * pred[:, -lfz:]
        index1 = min(S - 1, i + 1)
        latent = torch.cat([noise[:, :-lfz] * sigmas[index1] + (1 - sigmas[index1]) * model_input[:, :-lfz], temp_x0], dim=1)
    return latent


def tts(transformer, latent, model_input, noise, sigmas, lfz, randn, sde=True, guide=5.0, cfg=True, renoise=True):
    """sample_tts.py:694-868. cfg/renoise=False give the BASELINE config-4 composition (5B model: no CFG, clean history)."""
    S = len(sigmas)
    time_travel_step, time_travel_interval = 2, 2
    current_pred = None

    def hist(idx):
        if renoise:
            return noise[:, :-lfz] * sigmas[idx] + (1 - sigmas[idx]) * model_input[:, :-lfz]
        return model_input[:, :-lfz]

    def velocity(lat, idx):
        c = transformer(lat, idx, "cond")
        if not cfg:
            return c
        u = transformer(lat, idx, "uncond")
        return u + guide * (c - u)

    for i in range(S):
        pred = velocity(latent, i)
        nxt = 0 if i + 1 == S else sigmas[i + 1]
        temp_x0 = latent[:, -lfz:] + (nxt - sigmas[i]) * pred[:, -lfz:]
        if sde:
            prev_sample_mean = temp_x0
            pred_original_sample = latent[:, -lfz:] + (0 - sigmas[i]) * pred[:, -lfz:]
            eta = 0.3
            last50 = (i + 1 == 50)
            delta_t = 0 if last50 else (sigmas[i] - nxt)
            if delta_t < 0:
                delta_t = 0
            dsigma = (0 - sigmas[i]) if last50 else (nxt - sigmas[i])
            std_dev_t = eta * math.sqrt(delta_t)
            score_estimate = -(latent[:, -lfz:] - pred_original_sample * (1 - sigmas[i])) / sigmas[i] ** 2
            log_term = -0.5 * eta ** 2 * score_estimate
            prev_sample_mean = prev_sample_mean + log_term * dsigma
            temp_x0 = prev_sample_mean + randn(prev_sample_mean.shape) * std_dev_t
        if time_travel_interval > 0 and i % time_travel_interval == 0:
            travel_step = min(S - 1, i + time_travel_step)
            latent_travel = torch.cat([hist(travel_step), temp_x0], dim=1)
            for j in range(i + 1, travel_step):
                pt = velocity(latent_travel, j)
                temp_x0_travel = latent_travel[:, -lfz:] + (sigmas[j + 1] - sigmas[j]) * pt[:, -lfz:]
                if sde:
                    psm = temp_x0_travel
                    pos = latent_travel[:, -lfz:] + (0 - sigmas[j]) * pt[:, -lfz:]
                    eta = 0.3
                    delta_t = sigmas[j] - sigmas[j + 1]
                    if delta_t < 0:
                        delta_t = 0
                    dsigma = sigmas[j + 1] - sigmas[j]
                    std_dev_t = eta * math.sqrt(delta_t)
                    score = -(latent_travel[:, -lfz:] - pos * (1 - sigmas[j])) / sigmas[j] ** 2
                    psm = psm + (-0.5 * eta ** 2 * score) * dsigma
                    temp_x0_travel = psm + randn(psm.shape) * std_dev_t
                latent_travel = torch.cat([hist(min(S - 1, j + 1)), temp_x0_travel], dim=1)
                current_pred = pt
            temp_x0 = latent[:, -lfz:] + (nxt - sigmas[i]) * current_pred[:, -lfz:]
        latent = torch.cat([hist(min(S - 1, i + 1)), temp_x0], dim=1)
    return latent


def long_video_5b(transformer, vae_decode, first_history, n_chunks, sigmas, lfz, randn):
    """The FramePack chunk loop around euler_5b, fastvideo/sample/sample_5b.py:920-1097, as that script runs it for image/video-to-video
    input (`not t2v`), restated by reading (the hand-over lines sit between dataset, tokenizer and mp4 side effects and cannot be cut out
    and executed the way the inner loop is):

      chunk k (`step_sample`):  the conditioning latents `model_input` so far are padded with `lfz` frames (:1093-1095 `model_input_1`;
                                for the first chunk wan_i2v.generate hands back the same thing), fresh noise of that padded shape is drawn
                                (:945 `torch.randn_like(model_input_1)`) and only its last lfz frames are used (:953): latent = [history | noise];
                                Euler steps with the CLEAN history put back in front after every step (:1031-1034) = euler_5b;
                                the lfz new frames are appended to the conditioning latents (:1043-1049) and decoded on their own
                                (:1051-1052 `scale(vae, model_input[:, -lfz:])` -> vae.decode of those frames in fp32, :479-489).

    transformer(latent, i, "cond", k): velocity for chunk k's caption; vae_decode(z) -> video or None; randn(shape) replays the noise.
    Returns (all latents [C, F0 + n_chunks * lfz, H, W], [decoded chunk, ...])."""
    model_input = first_history
    videos = []
    for k in range(n_chunks):
        C, _, H, W = model_input.shape
        model_input_1 = torch.cat([model_input, torch.zeros(C, lfz, H, W, dtype=model_input.dtype, device=model_input.device)], dim=1)
        noise = randn(model_input_1.shape)
        latent = torch.cat([model_input_1[:, :-lfz], noise[:, -lfz:]], dim=1)
        latent = euler_5b(lambda lat, i, which: transformer(lat, i, which, k), latent, model_input_1, sigmas, lfz)
        model_input = torch.cat([model_input, latent[:, -lfz:]], dim=1)
        if vae_decode is not None:
            videos.append(vae_decode(model_input[:, -lfz:].float()))
    return model_input, videos
