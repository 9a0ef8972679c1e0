"""Post-decode frame path: decoder output fp32 [C,T,H,W] in [-1,1] -> uint8 frames.

Mirrors the one diffusers class the reference's drivers use after `vae.decode`
(`fastvideo/sample/sample_5b.py:491-500`: `VideoProcessor(vae_scale_factor=8).postprocess_video(video.unsqueeze(0),
output_type="pil")`, diffusers==0.32.0 per the reference's requirements.txt:27): same constructor argument, same method
name and argument meaning. The conversion itself runs in HBM through `yume_frames_u8` (include/yume_hip.h) and is
bit-identical to the host arithmetic diffusers does (`(x*0.5+0.5).clamp(0,1)` then `(x*255).round().astype(uint8)`);
only the uint8 frames cross PCIe. Encoding to mp4 (`export_to_video`, imageio/ffmpeg) stays host glue and is not here.
"""
import torch

from . import _lib


def frames_u8(video):
    """video fp32 [C,T,H,W] (device, contiguous) -> uint8 [T,H,W,C] (device)."""
    if not isinstance(video, torch.Tensor) or video.device.type != "cuda":
        raise RuntimeError("yume_amd.video: the video must be a device ('cuda') tensor — this path has no CPU fallback")
    if video.dim() != 4 or video.dtype != torch.float32:
        raise RuntimeError(f"yume_amd.video: expected fp32 [C,T,H,W], got {video.dtype} {tuple(video.shape)}")
    video = video.contiguous()
    C, T, H, W = video.shape
    out = torch.empty((T, H, W, C), dtype=torch.uint8, device=video.device)
    rc = _lib.load().yume_frames_u8(video.data_ptr(), C, T, H, W, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "yume_frames_u8")
    return out


class VideoProcessor:
    """Drop-in for the subset of diffusers.video_processor.VideoProcessor the reference calls."""

    def __init__(self, do_resize=True, vae_scale_factor=8, **_unused):
        self.vae_scale_factor = vae_scale_factor

    def postprocess_video(self, video, output_type="np"):
        """video [B,C,T,H,W] fp32 on the device. output_type: "uint8" -> torch uint8 [B,T,H,W,C] on the device (native);
        "np" -> float32 numpy [B,T,H,W,C] in [0,1]; "pt" -> torch [B,T,C,H,W] in [0,1]; "pil" -> list (batch) of lists of
        PIL images, as diffusers returns them."""
        if video.dim() != 5:
            raise ValueError(f"postprocess_video expects [B,C,T,H,W], got {tuple(video.shape)}")
        if output_type in ("uint8", "pil"):
            frames = torch.stack([frames_u8(v.float()) for v in video])
            if output_type == "uint8":
                return frames
            from PIL import Image
            host = frames.cpu().numpy()
            return [[Image.fromarray(f) for f in vid] for vid in host]
        den = (video.float() * 0.5 + 0.5).clamp(0, 1)          # diffusers VaeImageProcessor.denormalize
        if output_type == "pt":
            return den.permute(0, 2, 1, 3, 4)
        if output_type == "np":
            return den.permute(0, 2, 3, 4, 1).cpu().numpy()
        raise ValueError(f"unsupported output_type {output_type!r}")
