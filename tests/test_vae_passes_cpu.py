"""The VAE engine runs the reference's chunks several at a time (yume_amd/vae.py: decode_passes / encode_passes). The schedule must cover
exactly the frames the reference's loops cover (wan23/modules/vae2_2.py:802-820 encode, :839-857 decode), in order, first chunk alone."""
import pytest

from yume_amd.vae import decode_passes, encode_passes


def ref_decode_chunks(T):            # for i in range(iter_): decoder(x[:, :, i:i+1], first_chunk = (i == 0))
    return [(i, i + 1) for i in range(T)]


def ref_encode_chunks(T):            # i == 0: x[:, :, :1]; else x[:, :, 1 + 4*(i-1) : 1 + 4*i], i < 1 + (T-1)//4
    return [(0, 1)] + [(1 + 4 * (i - 1), 1 + 4 * i) for i in range(1, 1 + (T - 1) // 4)]


@pytest.mark.parametrize("group", [1, 2, 3, 7, 8, 100])
@pytest.mark.parametrize("T", [1, 2, 3, 8, 9, 13, 17, 21])
def test_decode_passes_cover_the_reference_walk(T, group):
    passes = decode_passes(T, group)
    assert passes[0] == (0, 1, True) and all(not f for _, _, f in passes[1:])
    assert [x for a, b, _ in passes for x in range(a, b)] == [x for a, b in ref_decode_chunks(T) for x in range(a, b)]
    assert all(0 < b - a <= max(1, group) for a, b, _ in passes)
    if group == 1:
        assert [(a, b) for a, b, _ in passes] == ref_decode_chunks(T)


@pytest.mark.parametrize("group", [1, 2, 3, 8, 100])
@pytest.mark.parametrize("T", [1, 2, 4, 5, 6, 9, 17, 29, 32, 33, 49])
def test_encode_passes_cover_the_reference_chunks(T, group):
    passes = encode_passes(T, group)
    ref = ref_encode_chunks(T)
    assert passes[0] == (0, 1, True) and all(not f for _, _, f in passes[1:])
    assert [x for a, b, _ in passes for x in range(a, b)] == [x for a, b in ref for x in range(a, b)]
    # a pass is a whole number of the reference's 4-frame chunks and starts on a chunk boundary
    assert all((b - a) % 4 == 0 and (a - 1) % 4 == 0 and 0 < (b - a) // 4 <= group for a, b, _ in passes[1:])
    if group == 1:
        assert [(a, b) for a, b, _ in passes] == ref
