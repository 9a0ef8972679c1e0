"""TEST INFRASTRUCTURE — golden vectors for the width-tiled decode helper, produced by the REFERENCE's own function.

webapp_single_gpu.py cannot be imported (it needs flask, diffusers, ... and loads checkpoints at import), so the one function
is lifted out of it with `ast` at generation time and executed as is, against a deterministic stand-in for `vae.decode`
(tests/test_frames.py::FakeVae restates it). Nothing from the reference is stored in the repo: only the inputs and outputs.
Run here (the container with /root/reference):  python oracle/make_golden_tiled.py
"""
import ast
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/webapp_single_gpu.py"


class FakeVae:
    """decode([z [C,T,H,W]]) -> [img [3, 4(T-1)+1, 16H, 16W]]: nearest upsampling of z's first three channels plus a term that
    depends on the whole band (its mean), so that a band decoded alone differs from the same columns of a wider band."""

    def decode(self, zs):
        z = zs[0].float()
        t = 4 * (z.shape[1] - 1) + 1
        up = z[:3].repeat_interleave(16, dim=2).repeat_interleave(16, dim=3)
        up = up[:, torch.arange(t) // 4 if z.shape[1] > 1 else torch.zeros(t, dtype=torch.long)]
        ramp = torch.linspace(-0.25, 0.25, up.shape[3]).view(1, 1, 1, -1)
        return [up + z.mean() * 0.5 + ramp]


def reference_function():
    tree = ast.parse(open(REF).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "tiled_decode_overlap")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    return ns["tiled_decode_overlap"]


def main():
    ref = reference_function()
    g = torch.Generator().manual_seed(11)
    cases = []
    for (c, t, h, w, n_tiles, ov, lfz) in [(4, 2, 1, 20, 5, 32, None), (4, 3, 1, 23, 5, 32, 2), (3, 1, 1, 12, 3, 16, None),
                                           (4, 2, 2, 17, 4, 48, 1), (4, 1, 1, 80, 5, 32, None)]:
        z = torch.randn(c, t, h, w, generator=g)
        with contextlib.redirect_stdout(io.StringIO()):
            out = ref(FakeVae(), z, n_tiles=n_tiles, image_overlap_size=ov, latent_frame_zero=lfz)
        cases.append(dict(z=z, n_tiles=n_tiles, image_overlap_size=ov, latent_frame_zero=lfz, out=out.clone()))
    path = os.path.join(ROOT, "tests", "golden", "tiled_decode.pt")
    torch.save(cases, path)
    print("wrote", path, [tuple(cs["out"].shape) for cs in cases])


if __name__ == "__main__":
    sys.exit(main())
