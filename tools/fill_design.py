#!/usr/bin/env python3
"""fill_design.py <bench.json-line file>: writes the round's measured numbers from bench.py's JSON line into the <<...>> placeholders of DESIGN.md
(section 3's "in the bench" column, section 4). Run once per round on the committed profiles/rN_bench_*.log."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
src = sys.argv[1]
g = {r["group"]: r for r in [d["roofline"]] + d["roofline_all"]}


def pf(name):
    r = g[name]
    return f"{r['launch_ms'] * 1e3:.0f} µs, {r['achieved'] / 1e3:.2f} PF = {r['frac']:.3f}"


s = open("DESIGN.md").read()
a = g["attn_self"]
s = s.replace("⟨ATTN8⟩", f"{a['launch_ms']:.4f} ms per 5B launch (kernel + merge pass) = {a['achieved'] / 1e3:.3f} PF = **{a['frac']:.3f}** of peak; {a['ms_per_step']:.1f} ms = "
              f"{a['share_of_step'] * 100:.0f} % of the step; fabric traffic {a['traffic'] / 1e6:.0f} MB vs 232 MB algorithmic ({a['traffic'] / 232.5e6:.2f} ×; `profiles/r4_pmc_traffic_attention_v8.csv`: "
              "MFMA busy 66.9 % at 1.75 GHz on random data)")
x = g["attn_cross"]
s = s.replace("⟨ATTNX⟩", f"{pf('attn_cross')}; {x['ms_per_step']:.1f} ms per step" + (f"; fabric traffic {x['traffic'] / 1e6:.0f} MB vs 65 MB algorithmic" if x.get("traffic") else ""))
s = s.replace("⟨GEMM⟩", "; ".join(f"{n[5:]} {pf(n)}" for n in ("gemm_qkv", "gemm_ffn0", "gemm_ffn2", "gemm_cross_q", "gemm_o", "gemm_cross_o")))
v = d.get("vae_decode") or {}
s = s.replace("⟨CONV⟩", f"Wan2.2 chunk decode {v.get('ms_per_chunk', float('nan')):.1f} ms = {v.get('latents_per_s', float('nan')):.1f} latents/s = {v.get('tflops', float('nan')) / 1e3:.2f} PF average")
s = s.replace("⟨GLUE⟩", f"adaLN {g['adaln']['launch_ms'] * 1e3:.1f} µs = {g['adaln']['achieved'] / 1e3:.2f} TB/s = {g['adaln']['frac']:.2f}; RMSNorm+RoPE {g['rmsnorm_rope']['launch_ms'] * 1e3:.1f} µs")
w = d.get("workloads", {})
cb, par = d.get("cpu_baseline", {}), d.get("parity", {})
fs = par.get("full_step", {})
lines = [
    f"`{src}` (the default `python bench.py`, one box; boxes differ by ±2–3 %): **{d['ms_per_step']:.2f} ms per step = {d['value']:.2f} denoise-steps/s, "
    f"{d['model_tflops_per_gpu'] / 1e3:.3f} PFLOP/s model level = {d['model_tflops_per_gpu'] / 2500:.3f} of nominal peak** (118.8 TFLOP per step; r3 96.1–101.7 ms, r2 105.3, r1 116.4). "
    f"With `cache_context`: {d.get('cached_context_ms_per_step', float('nan')):.2f} ms. The same command under `rocprofv3 --kernel-trace --stats`: "
    "`profiles/r4_bench_rocprofv3_kernel_stats.csv` (`attn_fwd_kernel_v8` 867.8 µs average over 480 launches + 9.9 µs merge pass; own kernels > 90 % of the profiled time).",
    "",
    "| group | launches/step | per launch | achieved | of peak | ms/step |",
    "|---|---|---|---|---|---|",
]
for r in [d["roofline"]] + d["roofline_all"]:
    ach = f"{r['achieved'] / 1e3:.3f} " + ("PF" if r.get("unit") == "TFLOP/s" else "TB/s") if r.get("achieved") else "–"
    lines.append(f"| {r['group']} | {r['launches_timed'] / (d['steps'] if r is d['roofline'] else 2):.0f} | {r['launch_ms'] * 1e3:.1f} µs | {ach} | {r.get('frac', float('nan')):.3f} | {r['ms_per_step']:.2f} |")
lines += ["",
          f"VAE (Wan2.2, second half of the metric): {v.get('ms_per_chunk', float('nan')):.1f} ms per 8-latent 704×1280 chunk = **{v.get('latents_per_s', float('nan')):.1f} latents/s** = "
          f"{v.get('tflops', float('nan')) / 1e3:.2f} PF average = {v.get('tflops', float('nan')) / 2500:.2f} of peak; CPU oracle decoder {v.get('cpu_baseline', {}).get('value', float('nan')):.4f} latents/s on "
          f"{v.get('cpu_baseline', {}).get('cores', '?')} host threads.",
          "",
          f"CPU baseline (`cpu_baseline`, kind \"port\": `oracle/dit.py`, pinned to the reference): **ONE WHOLE denoise step measured** — {1.0 / cb['value']:.0f} s = {cb['value']:.5f} steps/s on "
          f"{cb['cores']} of {cb.get('host_threads', '?')} host threads (r3 and before: one block × 30); the block sample and its thread sweep stay in `cpu_baseline.block_sample`. "
          f"GPU / CPU = {d['value'] / cb['value']:.0f} × (a reported baseline, not a target).",
          "",
          f"Parity in the same line: one live block rel-L2 {par.get('block', {}).get('rel_l2', float('nan')):.1e} (update {par.get('block', {}).get('update_rel_l2', float('nan')):.1e}); **the whole step** "
          f"(30 live blocks + head + Euler update vs the fp32 oracle): velocity rel-L2 {fs.get('pred_rel_l2', float('nan')):.2e}, max-abs {fs.get('pred_max_abs', float('nan')):.1e}; updated latent rel-L2 "
          f"{fs.get('latent_rel_l2', float('nan')):.1e} (tolerances 3e-2 / 2e-3, §5).",
          "",
          "The other single-GPU configurations ride in the same JSON line as `workloads` (each with its own barrier-bracketed timed region; `python bench.py --workload X` runs one alone, at any N):",
          ""]
if w:
    t, lv, b14 = w.get("tts", {}), w.get("longvideo", {}), w.get("14b", {})
    lines += [f"* configs[2] Yume-I2V-14B-540P, L = 27810, CFG: **{b14.get('ms_per_step', float('nan')):.0f} ms per CFG step** = {b14.get('model_tflops_per_gpu', float('nan')) / 1e3:.2f} PF model level (r3 2035).",
              f"* configs[3] 5B SDE/TTS: {t.get('ms_per_forward', float('nan')):.1f} ms per forward, {t.get('value', float('nan')):.2f} sampler steps/s ({t.get('forwards_per_50_step_chunk', 74)} forwards per 50-step chunk).",
              f"* configs[4] FramePack long video, 8 chunks × {lv.get('steps', 16) // 8} steps, L {lv.get('config', {}).get('tokens_per_chunk', ['?'])[0]} … {lv.get('config', {}).get('tokens_per_chunk', ['?'])[-1]}, VAE encode + 8 decodes inside "
              f"the timed region: {lv.get('value', float('nan')):.2f} steps/s, {lv.get('latents_per_s', float('nan')):.1f} latents/s; one chunk's parts: {lv.get('parts_of_one_chunk')}."]
lines += ["",
          "Multi-GPU: independent chains, no collective in the loop (§6); no 8-GPU lease was available to the builder — the driver's SCALE run is the measurement.",
          "PCIe: the boundary takes device tensors (as the reference's `transformer(latent, …)` does), so no host buffers cross per step."]
s = s.replace("⟨STEP⟩", "\n".join(lines))
open("DESIGN.md", "w").write(s)
print("DESIGN.md filled from", src)
