"""TEST INFRASTRUCTURE — runs the web app's own `_postprocess_video` (oracle/ref_scripts.py executes the function cut out of
webapp_single_gpu.py) on a seeded video and stores the frames as tests/golden/frames_webapp.pt.

    python oracle/make_golden_frames.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_scripts as R  # noqa: E402


def main():
    assert R.available(), "needs the reference tree"
    v = R.webapp_case()
    frames = R.run_webapp_postprocess(v.clone())
    path = os.path.join(ROOT, "tests", "golden", "frames_webapp.pt")
    torch.save(dict(shape=tuple(v.shape), checksum=float(v.double().sum()), frames=torch.from_numpy(frames)), path)
    print("wrote", path, os.path.getsize(path), "bytes", frames.shape, frames.dtype)


if __name__ == "__main__":
    main()
