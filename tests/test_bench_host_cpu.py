"""bench.py's host-side pieces that need no GPU: the committed evidence files its JSON line quotes are present and parse, the FLOP model
of the step matches SURVEY §8(d), and the roofline table carries `frac_of_sustained` when a calibration is attached."""
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_flop_model_of_the_5b_step():
    # 118.8 TFLOP per forward at L = 9460 (30 blocks, 512 padded text tokens)
    assert abs(bench.flops_fwd_5b(9460) / 1e12 - 118.8) < 0.1
    assert bench.block_flops_5b(9460, dict(dim=3072, ffn_dim=14336)) * 30 < bench.flops_fwd_5b(9460)


def test_pmc_traffic_comes_from_this_rounds_committed_pass():
    a = bench.pmc_traffic_bytes("attn_self")              # kernel + merge pass, 2 * FETCH + WRITE [KiB] of profiles/r5_pmc_traffic_attention_v8.csv
    x = bench.pmc_traffic_bytes("attn_cross")
    assert 4.0e8 < a < 6.0e8 and 1.0e8 < x < 2.0e8
    assert a > 232e6 and x > 65e6                         # never below the algorithmic bytes
    assert os.path.exists(os.path.join(ROOT, "profiles", bench.PMC_FILES[0]))
    g = bench.pmc_traffic_bytes("gemm_ffn0")              # per-shape pass (main launch + row remainder)
    assert g is not None and g > 417e6


def test_port_vs_reference_record_is_attached():
    r = bench.port_vs_reference()
    assert "failed" not in r
    assert 0.8 < r["reference_over_port"] < 1.3 and r["outputs_rel_l2"] <= 1e-6
    rec = json.load(open(os.path.join(ROOT, "profiles", "r5_cpu_port_vs_reference.json")))
    assert rec["reference_over_port"] == r["reference_over_port"]


def test_bench_defaults_are_the_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--steps", type=int, default=6' in src and '"--warmup", type=int, default=2' in src
    assert "MFMA_BF16_PEAK_TFLOPS = 2500.0" in src
    for key in ('"roofline"', '"cpu_baseline"', '"calibration"', '"vs_baseline": None', '"higher_is_better": True', '"scaling"'):
        assert key in src, key
