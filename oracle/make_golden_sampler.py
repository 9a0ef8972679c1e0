"""TEST INFRASTRUCTURE — runs the denoise loops of the reference's sampling scripts (oracle/ref_scripts.py cuts them out of the script
text and executes them) on small seeded inputs and stores inputs' recipe + outputs as tests/golden/sampler_scripts.pt, so that the GPU box
(no /root/reference) can still check oracle/sampler.py and yume_amd/sampling.py against what the scripts themselves compute.

    python oracle/make_golden_sampler.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_scripts as R  # noqa: E402
from oracle import sampler as S  # noqa: E402

LFZ, SEED_CASE, SEED_FIELD, SEED_NOISE = 3, 11, 5, 7


def main():
    assert R.available(), "needs the reference tree"
    out = {"lfz": LFZ, "seed_case": SEED_CASE, "seed_field": SEED_FIELD, "seed_noise": SEED_NOISE, "cases": {}}
    for dtype in (torch.float64, torch.float32):
        c = R.script_case(SEED_CASE, dtype)
        f = R.script_field(SEED_FIELD, c["C"])
        mi, noise = c["model_input"], c["noise"]
        key = str(dtype).split(".")[-1]
        res = {}
        sig3 = list(S.get_sampling_sigmas(50, 3.0))
        lat, draws, calls, span = R.run_tts(f, noise.clone(), noise, mi, sig3, LFZ, sde=True, seed=SEED_NOISE)
        res["tts_50"] = dict(latent=lat, n_calls=len(calls), n_draws=len(draws), lines=span)
        lat, draws, calls, span = R.run_tts(f, noise.clone(), noise, mi, sig3, LFZ, sde=False, seed=SEED_NOISE)
        res["tts_50_ode"] = dict(latent=lat, n_calls=len(calls), n_draws=len(draws), lines=span)
        for n in (50, 6):
            sig = list(S.get_sampling_sigmas(n, 3.0))
            lat, calls, span = R.run_euler_14b(f, noise.clone(), noise, mi, sig, LFZ)
            res[f"euler14b_{n}"] = dict(latent=lat, n_calls=len(calls), lines=span)
            sig = list(S.get_sampling_sigmas(n, 7.0))
            seq_len = c["F"] * (c["H"] // 2) * (c["W"] // 2) + 5        # (the script pads t up to arg_c['seq_len'])
            lat0 = torch.cat([mi[:, :-LFZ], noise[:, -LFZ:]], dim=1)
            lat, tvecs, calls, span = R.run_euler_5b(f, lat0, mi, sig, LFZ, seq_len)
            res[f"euler5b_{n}"] = dict(latent=lat, n_calls=len(calls), lines=span, seq_len=seq_len, t=torch.stack(tvecs))
        out["cases"][key] = res
    path = os.path.join(ROOT, "tests", "golden", "sampler_scripts.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
