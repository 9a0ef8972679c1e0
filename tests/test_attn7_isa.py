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
    src = os.path.join(build.CSRC, "attn_fwd7.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "a7.s")
        cmd = [build._hipcc()] + build.FLAGS + build.EXTRA_FLAGS.get("attn_fwd7.hip", []) + \
              ["-I", build.INCLUDE, "--cuda-device-only", "-S", src, "-o", out]
        subprocess.run(cmd, check=True, capture_output=True, text=True)
        txt = open(out).read()
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
            lo = int(m.group(1))
            bad.append(t)        # the kernel owns every AGPR: hipcc may use none
    assert not bad, "compiler-generated code touches the AGPRs the kernel owns:\n" + "\n".join(bad[:10])
    # the steady loop (4 key tiles per trip): 256 MFMAs, its own four counted waits and nothing else that waits on memory —
    # no spill traffic, no compiler-inserted s_waitcnt vmcnt(0) (it would also wait for every LDS-DMA piece in flight)
    lines = body.split("\n")
    hdr = next(i for i, l in enumerate(lines) if "Inner Loop Header: Depth=2" in l)
    label = next(lines[j].split(":")[0].strip() for j in range(hdr, hdr - 4, -1) if lines[j].startswith(".LBB"))
    end = next(i for i in range(hdr, len(lines)) if ("s_cbranch" in lines[i] or "s_branch" in lines[i]) and lines[i].split()[-1] == label)
    loop = lines[hdr:end]
    assert sum("v_mfma_f32_32x32x16_bf16" in l for l in loop) == 256
    assert sum("s_waitcnt vmcnt(8)" in l for l in loop) == 4
    assert not [l for l in loop if "scratch_" in l or ("vmcnt(" in l and "vmcnt(8)" not in l)], "memory waits / spills inside the steady loop"
    assert sum("ds_read_b128" in l for l in loop) == 4 * 48          # 16 K fragments (once, into AGPRs) + 32 V^T fragments per tile
