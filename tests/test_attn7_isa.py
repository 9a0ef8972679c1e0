"""attn_fwd7.hip names all 256 AGPRs (O^T and Q^T of its two query blocks, the K fragment cache) literally in inline asm. hipcc must keep nothing of its
own in those registers: compile the file to assembly (a few seconds, no GPU) and check that no compiler-generated instruction
— anything outside the ;;#ASMSTART/;;#ASMEND brackets — touches them, and that the kernel neither spills nor uses scratch."""
import os
import re
import subprocess
import tempfile

from conftest import ROOT


def test_owned_agprs_are_untouched_by_the_compiler():
    from yume_amd import build
    txt = build.device_asm("attn_fwd7.hip")          # the assembly of the library's own build (kept next to the objects)
    body = txt[txt.index("attn_fwd_kernel_v7"):]
    inasm, bad = False, []
    for line in body.split("\n"):
        if "#ASMSTART" in line:
            inasm = True
            continue
        if "#ASMEND" in line:
            inasm = False
            continue
        t = line.strip()
        if inasm or not t or t[0] in ";.":
            continue
        for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", t.split(";")[0]):
            bad.append(t)        # the kernel owns every AGPR: hipcc may use none
    assert not bad, "compiler-generated code touches the AGPRs the kernel owns:\n" + "\n".join(bad[:10])
    # The steady loops (4 key tiles per trip): one in the plain kernel, two in the YUME_ATTN_Q_PRESCALED one (its base-free pieces and
    # the robust rerun). Each: 256 MFMAs, its own four counted waits and nothing else that waits on memory — no spill traffic, no
    # compiler-inserted s_waitcnt vmcnt(N) (it would also wait for the LDS-DMA pieces in flight); 16 K fragments (once, into AGPRs) +
    # 32 V^T fragments read per tile; one wait per PAIR of V^T fragments; the base-free loop has no shift (v_fma) and no range vote.
    lines = body.split("\n")
    hdrs = [i for i, l in enumerate(lines) if "Inner Loop Header: Depth=2" in l]
    assert len(hdrs) == 3
    n_fast = 0
    for hdr in hdrs:
        label = next(lines[j].split(":")[0].strip() for j in range(hdr, hdr - 4, -1) if lines[j].startswith(".LBB"))
        end = next(i for i in range(hdr, len(lines)) if ("s_cbranch" in lines[i] or "s_branch" in lines[i]) and lines[i].split()[-1] == label)
        loop = [l for l in lines[hdr:end] if l.strip() and l.strip()[0] not in ";."]
        assert sum("v_mfma_f32_32x32x16_bf16" in l for l in loop) == 256
        assert sum("s_waitcnt vmcnt(8)" in l for l in loop) == 4
        assert not [l for l in loop if "scratch_" in l or ("vmcnt(" in l and "vmcnt(8)" not in l)], "memory waits / spills inside the steady loop"
        assert sum("ds_read_b128" in l for l in loop) == 4 * 48
        assert sum("global_load_lds_dwordx4" in l for l in loop) == 4 * 8
        assert sum("s_waitcnt lgkmcnt" in l for l in loop) <= 4 * 16
        assert sum("v_exp_f32" in l for l in loop) == 256
        nshift = sum(l.split()[0].startswith(("v_fma_f32", "v_sub_f32", "v_subrev_f32")) for l in loop)     # (scale 1 folds the fma into a subtraction)
        assert nshift in (0, 256)
        if nshift == 0:
            n_fast += 1
            assert not [l for l in loop if "s_cbranch" in l or "v_max3_f32" in l], "the base-free loop carries a range vote"
            assert len(loop) <= 5.5 * 256, f"{len(loop)} instructions for 256 MFMA gaps"
    assert n_fast == 1
