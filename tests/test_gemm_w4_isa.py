"""gemm_w4.hpp / conv_w4.hpp keep all 256 accumulator AGPRs of a wave to themselves (named literally in inline asm, listed as clobbers).
hipcc must keep nothing of its own there — r3 found it spilling THROUGH a[0:3] when several 128-MFMA loop bodies met at a join — so:
compile the two translation units to assembly (no GPU) and check, for the one-wave-per-SIMD kernels, that no compiler-generated
instruction (anything outside ;;#ASMSTART / ;;#ASMEND) touches an AGPR, that nothing spills or uses scratch, and that each steady K loop
is exactly the gap plan: 128 MFMAs, 32 fragment reads, 16 LDS-DMA pieces, 2 barriers, no memory wait the plan does not own."""
import re

import pytest

from conftest import ROOT


def _asm(src_name):
    from yume_amd import build
    return build.device_asm(src_name)          # the assembly of the library's own build (kept next to the objects)


def _kernel_body(txt, mangled_prefix):
    lines = txt.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel " + mangled_prefix in lines[i])
    meta = "\n".join(lines[end:end + 120])
    return lines[start:end], meta


def _compiler_agpr_uses(body):
    inasm, bad = False, []
    for line in body:
        if "#ASMSTART" in line:
            inasm = True
            continue
        if "#ASMEND" in line:
            inasm = False
            continue
        t = line.strip()
        if inasm or not t or t[0] in ";.":
            continue
        if "v_accvgpr" in t or re.search(r"\ba\[?\d+", t.split(";")[0]):
            bad.append(t)
    return bad


def _steady_loops(body):
    """[(instructions)] of every backward-branch loop that holds 128 MFMAs and stages (a DMA inside)."""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            blk = [x.strip() for x in body[labels[m.group(1)]:i + 1] if x.strip() and x.strip()[0] not in ";."]
            n_mfma = sum("v_mfma_f32_16x16x32_bf16" in x for x in blk)
            if n_mfma == 128 and any("_load_lds_dwordx4" in x or ("buffer_load_dwordx4" in x and " lds" in x) for x in blk):
                loops.append(blk)
    return loops


# dense: one steady loop per operand order; conv: plain and folded-upsample sources (the walk of the taps — SALU, a branch when the frame
# changes — sits between two trips: more instructions per trip, the same plan)
@pytest.mark.parametrize("src,kernel,n_loops,budget", [("gemm_bf16.hip", "_ZN7gemm_w414gemm_w4_kernelILi0EE", 2, 2.3),
                                                       ("conv3d.hip", "_ZN7gemm_w414conv_w4_kernelILi0EE", 2, 3.2)])
def test_w4_kernels_own_their_accumulators_and_follow_the_gap_plan(src, kernel, n_loops, budget):
    txt = _asm(src)
    body, meta = _kernel_body(txt, kernel)
    bad = _compiler_agpr_uses(body)
    assert not bad, "compiler-generated code touches the AGPRs the kernel owns:\n" + "\n".join(bad[:10])
    assert not [l for l in body if "scratch_" in l], "the kernel uses scratch"
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta)
    loops = _steady_loops(body)
    assert len(loops) == n_loops, [len(b) for b in loops]          # operand orders (dense) / plain + folded-upsample sources (conv)
    for blk in loops:
        assert sum("ds_read_b128" in x for x in blk) == 32
        dma = [x for x in blk if "_load_lds_dwordx4" in x or ("buffer_load_dwordx4" in x and " lds" in x)]
        assert len(dma) == 16
        assert sum(x.startswith("s_barrier") for x in blk) == 2
        vm = [x for x in blk if "vmcnt(" in x]
        assert len(vm) == 1 and "lgkmcnt" not in vm[0], vm          # the one counted wait of the plan; hipcc adds none (it would drain the DMA in flight)
        assert not [x for x in blk if x.startswith("v_accvgpr")] and sum(x.startswith("v_mov_b32") for x in blk) <= 4, "register shuffling inside the steady loop"
        assert len(blk) <= budget * 128, f"{len(blk)} instructions for 128 MFMAs"


def test_w4_bf16_epilogue_stays_lean():
    """What the per-workgroup trace of r3 paid for (profiles/r3_gemm_w4.md section 7): the accumulator -> LDS image pass of a bf16 tile is
    ~10 instructions per 16x16 accumulator tile (4 v_accvgpr_read, 2 packed adds, 2 packed converts, 1 ds_write_b64 with the row-block offset
    in its offset field) — no per-tile address arithmetic — and the image -> global pass keeps 8 image reads in flight ahead of their stores
    instead of read / wait / 64-bit multiply / store per row."""
    body, _ = _kernel_body(_asm("gemm_bf16.hip"), "_ZN7gemm_w414gemm_w4_kernelILi0EE")
    ins = [x.strip() for x in body if x.strip() and x.strip()[0] not in ";." and not x.strip().endswith(":")]
    w = [i for i, x in enumerate(ins) if x.startswith("ds_write_b64")]
    runs, cur = [], [w[0]]
    for i in w[1:]:
        if i - cur[-1] > 60:
            runs.append(cur)
            cur = [i]
        else:
            cur.append(i)
    runs.append(cur)
    images = [r for r in runs if len(r) == 64]                       # one per bf16-output epilogue (plain, GELU, erf-GELU, q|k of SPLITT, V^T)
    assert len(images) >= 4
    plain = [r for r in images if not any(op.startswith(("v_exp_f32", "v_rcp_f32", "v_fma")) for op in ins[r[0]:r[-1]])]
    assert plain, "no plain bf16 image pass found"
    for r in plain:
        span = ins[r[0] - 8:r[-1] + 1]
        assert len(span) <= 64 * 12.5, f"{len(span)} instructions for 64 accumulator tiles"
        assert not [x for x in span if x.startswith(("v_mul_lo", "v_mad_u64", "v_lshl_add_u64"))], "per-tile address arithmetic is back"
    # image -> global: somewhere 8 ds_read_b128 stand in a row (no store between them), followed by global_store_dwordx4
    mem = [x.split()[0] for x in ins if x.startswith(("ds_read_b128", "global_store_dwordx4", "s_barrier", "v_mfma"))]
    best, run = 0, 0
    for i, op in enumerate(mem):
        run = run + 1 if op == "ds_read_b128" else 0
        if run >= 8 and i + 1 < len(mem) and mem[i + 1] == "global_store_dwordx4":
            best = max(best, run)
    assert best >= 8, "the image -> global pass no longer batches its reads"
