"""oracle/step_job.py on the build container: the subprocess plumbing of the full-depth parity legs (tests/test_zz_full_step_gpu.py,
bench.py's `parity.full_step`) on 2-layer models, and the lazily generated (hashed) state_dict against a materialised one."""
import os
import sys
import tempfile

import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle import dit as odit  # noqa: E402
from oracle import step_job  # noqa: E402
from yume_amd import synth  # noqa: E402


def test_subprocess_job_equals_in_process_forward_and_materialised_weights():
    name = "tiny5b"
    out = os.path.join(tempfile.gettempdir(), f"yume_step_test_{os.getpid()}.pt")
    proc = step_job.start_job(name, "cond", out, threads=2)
    try:
        pred, secs, gen = step_job.oracle_forward(name, "cond", threads=2)
        ref = step_job.finish_job(proc, out, timeout=600)
    finally:
        for f in (out, out + ".log"):
            if os.path.exists(f):
                os.remove(f)
    assert ref["case"] == name and ref["seconds"] > 0 and ref["threads"] == 2
    assert torch.allclose(ref["pred"], pred, rtol=0, atol=1e-5)          # two processes, same arithmetic (thread-count-dependent sums)
    # the same forward from a materialised dict of the lazily generated tensors, through the plain oracle entry point
    c, cfg = step_job.CASES[name], step_job.case_cfg(name)
    lazy = synth.HashedDitStateDict(cfg, "wan23", step_job.SEED)
    sd = {k: lazy[k] for k in lazy.keys()}
    inp, plan, sg = step_job.make_inputs(name), step_job.seq_len(name), step_job.sigmas(name)
    t = torch.cat([torch.zeros(plan.n_hist_tok, dtype=torch.float64), torch.full((plan.n_new_tok,), sg[c["i"]] * 1000.0, dtype=torch.float64)]).unsqueeze(0)
    want = odit.forward_wan23(sd, cfg, inp["latent"], t, inp["cond"], plan.seq_len, c["lfz"], True)
    assert torch.allclose(pred, want, rtol=0, atol=2e-5)                 # attention_fp32 (head-by-head fp32) vs the fp64 score matrix
    assert pred.shape == (cfg["out_dim"], c["lfz"], c["H"], c["W"]) and pred.abs().max() > 1e-3


def test_cfg_case_runs_both_contexts_and_euler_update():
    name = "tiny14b"
    c = step_job.CASES[name]
    pc, _, _ = step_job.oracle_forward(name, "cond", threads=2)
    pu, _, _ = step_job.oracle_forward(name, "uncond", threads=2)
    assert pc.shape == pu.shape and not torch.equal(pc, pu)
    lat = step_job.make_inputs(name)["latent"]
    v = pu + c["guide"] * (pc - pu)
    sg = step_job.sigmas(name)
    x = step_job.euler(name, lat, v, c["i"])
    assert torch.allclose(x, lat[:, -c["lfz"]:] + (sg[c["i"] + 1] - sg[c["i"]]) * v[:, -c["lfz"]:])
    assert x.shape == (16, c["lfz"], c["H"], c["W"])
