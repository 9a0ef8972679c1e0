"""attn_fwd8.hip (the persistent form of the one-wave-per-SIMD attention kernel) — the same audit as tests/test_attn7_isa.py plus what is
specific to the stream: hipcc must keep nothing of its own in the 256 AGPRs the kernel owns; the four-tile steady loop AND every
single-tile instance (steady, first behind a boundary, masked, item boundary: 4 slot phases each) carry at most their one counted wait per
tile (the first tile behind a boundary: none) and no spill traffic —
a compiler-inserted s_waitcnt vmcnt(0) inside a tile would drain the LDS-DMA queue (hipcc cannot count the pieces issued in inline asm)."""
import re

from conftest import ROOT  # noqa: F401


def _kernel_body():
    from yume_amd import build
    txt = build.device_asm("attn_fwd8.hip")
    body = txt[txt.index("attn_fwd_kernel_v8"):]
    end = body.find(".Lfunc_end")
    return body[:end if end > 0 else None]


def test_owned_agprs_are_untouched_by_the_compiler():
    inasm, bad = False, []
    for line in _kernel_body().split("\n"):
        if "#ASMSTART" in line:
            inasm = True
            continue
        if "#ASMEND" in line:
            inasm = False
            continue
        t = line.strip()
        if inasm or not t or t[0] in ";.":
            continue
        if re.search(r"\ba\[?(\d+)(?::(\d+))?\]?", t.split(";")[0]):
            bad.append(t)
    assert not bad, "compiler-generated code touches the AGPRs the kernel owns:\n" + "\n".join(bad[:10])


def _blocks():
    blocks, cur = [], []
    for l in _kernel_body().split("\n"):
        if l.startswith(".LBB"):
            blocks.append(cur)
            cur = []
        elif l.strip() and l.strip()[0] not in ";.":
            cur.append(l.strip())
    blocks.append(cur)
    return blocks


def test_every_stream_tile_has_one_counted_wait_and_no_spill_traffic():
    pat = re.compile(r"vmcnt\((\d+)\)")
    tiles = [b for b in _blocks() if sum("v_mfma_f32_32x32x16_bf16" in x for x in b) in (64, 256)]
    # base-free stream: 12 single tiles (3 kinds x 4 slot phases) + the four-tile loop; the robust rerun (attn_fwd7's run_keys) brings its
    # own blocks, recognised by their per-score shift (v_fma / v_sub) and left to tests/test_attn7_isa.py
    stream = [b for b in tiles if not any(x.split()[0].startswith(("v_fma_f32", "v_sub_f32", "v_subrev_f32")) for x in b)]
    singles = [b for b in stream if sum("v_mfma" in x for x in b) == 64]
    loops = [b for b in stream if sum("v_mfma" in x for x in b) == 256]
    # 16 single tiles: steady, first-behind-a-boundary, masked, boundary — 4 slot phases each
    assert len(singles) == 16 and len(loops) == 1, (len(singles), len(loops))
    unwaited = 0
    for b in singles + loops:
        n = sum("v_mfma" in x for x in b) // 64
        waits = [pat.search(x).group(1) for x in b if "vmcnt(" in x]
        if n == 1 and not waits:
            unwaited += 1               # the first tile behind an item boundary: barrier only (its inputs were waited for in bubble 1; a counted
            continue_checks = True      # wait would sit out the O^T stores of the item just finished)
        else:
            assert waits == ["8"] * n, waits
        assert not [x for x in b if "scratch_" in x], "spill traffic inside a stream tile"
        assert sum("global_load_lds_dwordx4" in x for x in b) == 8 * n
        assert sum("ds_read_b128" in x for x in b) == 48 * n
        assert sum("v_exp_f32" in x for x in b) == 64 * n
        assert sum("s_barrier" in x for x in b) == n
    assert unwaited == 4
    assert len(loops[0]) <= 5.6 * 256, f"{len(loops[0])} instructions for 256 MFMA gaps"
    masked = [b for b in singles if any("v_cndmask_b32" in x for x in b)]
    assert len(masked) == 8                      # the tile before an item's last and the boundary tile carry the key mask, the steady tile none
