"""attn_fwd7.hip names a[0:191] (O^T and Q^T of its two query blocks) literally in inline asm. hipcc must keep nothing of its
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
            if lo < 192:
                bad.append(t)
    assert not bad, "compiler-generated code touches the AGPRs the kernel owns:\n" + "\n".join(bad[:10])
    assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", txt).group(1)) == 0
    assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", txt).group(1)) == 0
    # the steady loop keeps its shape: 64 MFMAs per key tile between two counted waits
    assert txt.count("s_waitcnt vmcnt(8)") == 4
