"""TEST INFRASTRUCTURE — CPU restatement of the reference's post-decode frame conversion.

The reference (fastvideo/sample/sample_5b.py:491-500, :1068) hands the decoded video to a third-party dependency that is
NOT vendored under /root/reference: diffusers==0.32.0 (requirements.txt:27), `VideoProcessor.postprocess_video(...,
output_type="pil")`. Its published algorithm, restated:
    video_processor.py  postprocess_video: for each batch item, frames = video[b].permute(1, 0, 2, 3)  -> [T,C,H,W]
    image_processor.py  denormalize:       (images * 0.5 + 0.5).clamp(0, 1)
                        pt_to_numpy:       images.cpu().permute(0, 2, 3, 1).float().numpy()           -> [T,H,W,C]
                        numpy_to_pil:      (images * 255).round().astype("uint8")  (numpy: round half to even)
Parity pin: diffusers is not installed in this image, so the restatement is pinned to known-answer values derived from
the formula above (tests/test_frames.py), not to an execution of the dependency.
Only tests/ may import this module.
"""
import torch


def frames_u8(video):
    """video: torch fp32 [C,T,H,W] -> numpy uint8 [T,H,W,C]."""
    den = (video.float() * 0.5 + 0.5).clamp(0, 1)
    arr = den.permute(1, 2, 3, 0).contiguous().numpy()
    return (arr * 255).round().astype("uint8")


def frames_u8_webapp(video):
    """the web app's own conversion (webapp_single_gpu.py:117-121, IN-TREE reference code): ((v.clamp(-1,1) + 1) / 2 * 255).byte() — the cast
    truncates. Unlike the diffusers variant above this one is PINNED: tests/test_frames.py runs the reference function itself
    (oracle/ref_scripts.py::run_webapp_postprocess) against it, and tests/golden/frames_webapp.pt holds its output for the GPU box.
    video: torch fp32 [C,T,H,W] -> numpy uint8 [T,H,W,C]."""
    v = video.float().clamp(-1, 1).add(1).div(2)
    return (v * 255).byte().permute(1, 2, 3, 0).contiguous().numpy()
