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


def frames_u8(video, truncate=False):
    """video fp32 [C,T,H,W] (device, contiguous) -> uint8 [T,H,W,C] (device). truncate=False: diffusers' rounding (the sampling scripts);
    truncate=True: the web app's own `_postprocess_video` (webapp_single_gpu.py:117-121), which casts with .byte()."""
    if not isinstance(video, torch.Tensor) or video.device.type != "cuda":
        raise RuntimeError("yume_amd.video: the video must be a device ('cuda') tensor — this path has no CPU fallback")
    if video.dim() != 4 or video.dtype != torch.float32:
        raise RuntimeError(f"yume_amd.video: expected fp32 [C,T,H,W], got {video.dtype} {tuple(video.shape)}")
    video = video.contiguous()
    C, T, H, W = video.shape
    out = torch.empty((T, H, W, C), dtype=torch.uint8, device=video.device)
    lib = _lib.load()
    fn = lib.yume_frames_u8_trunc if truncate else lib.yume_frames_u8
    rc = fn(video.data_ptr(), C, T, H, W, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "yume_frames_u8")
    return out


def postprocess_video_webapp(video):
    """The frame conversion of the reference's web app (`_postprocess_video`, webapp_single_gpu.py:117-121, without its mp4 export):
    video fp32 [C,F,H,W] in [-1,1] on the device -> list of F PIL images; `((v.clamp(-1,1) + 1) / 2 * 255).byte()` runs in HBM."""
    from PIL import Image
    return [Image.fromarray(f) for f in frames_u8(video.float(), truncate=True).cpu().numpy()]


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


def _tile_spans(width, n_tiles, overlap):
    """latent column span [start, end) of each tile: near-equal widths, widened by `overlap` columns towards each neighbour."""
    base, rem = divmod(width, n_tiles)
    spans, at = [], 0
    for i in range(n_tiles):
        w = base + (1 if i < rem else 0)
        lo = at - (overlap if i > 0 else 0)
        hi = at + w + (overlap if i < n_tiles - 1 else 0)
        spans.append((max(lo, 0), min(hi, width)))
        at += w
    return spans


def tiled_decode_overlap(vae, latents, n_tiles=5, image_overlap_size=32, latent_frame_zero=None):
    """Width-tiled VAE decode with blended seams: same arguments and arithmetic as the reference's memory workaround
    (webapp_single_gpu.py:370-551, called at :830 with the 5B VAE): latents [C,T,H,W] is cut into `n_tiles` column bands, each
    widened by image_overlap_size//16 latent columns towards its neighbours, decoded on its own (`vae.decode([band])[0]`), and
    the bands are averaged with weights 1 for the first / last band and a linear ramp over `image_overlap_size` pixels at
    both ends of the inner bands. `latent_frame_zero` keeps only the last that many latent frames.

    On MI355X a whole 704x1280 chunk decodes in one piece (288 GB HBM), and that is what the drivers should call — a band
    decoded without its neighbours' context differs from the full decode near the seams. This function exists so that code
    written against the webapp's helper runs unchanged and produces what the reference's helper produces.
    The blend is built as one weight row per band (no per-column host loop) and accumulated on the latents' device.
    """
    scale = 16
    c, t, h, w = latents.shape
    out_w = w * scale
    spans = _tile_spans(w, n_tiles, max(1, image_overlap_size // scale))
    result, total = None, None
    for i, (lo, hi) in enumerate(spans):
        band = latents[:, -latent_frame_zero:, :, lo:hi] if latent_frame_zero is not None else latents[:, :, :, lo:hi]
        img = vae.decode([band])[0]
        p0, p1 = lo * scale, min(hi * scale, out_w)
        pw = p1 - p0
        if result is None:
            result = torch.zeros(img.shape[0], img.shape[1], img.shape[2], out_w, device=img.device, dtype=img.dtype)
            total = torch.zeros(out_w, device=img.device, dtype=torch.float32)
        if i == 0 or i == n_tiles - 1:
            wrow = torch.ones(pw, device=img.device, dtype=torch.float32)
        else:
            j = torch.arange(pw, device=img.device, dtype=torch.float32)
            wrow = torch.where(j < image_overlap_size, j / image_overlap_size,
                               torch.where(j > pw - image_overlap_size, (pw - j) / image_overlap_size, torch.ones_like(j)))
        if img.shape[3] != pw:
            img = torch.nn.functional.interpolate(img, size=(img.shape[2], pw), mode="bilinear", align_corners=False)
        result[..., p0:p1] += img * wrow
        total[p0:p1] += wrow
    return result / total.clamp(min=1e-8)
