"""Checkpoint I/O next to the hot path: safetensors weights streamed tensor by tensor onto the GPU.

Reference: the Yume checkpoints are diffusers-style directories (config.json + diffusion_pytorch_model.safetensors, sharded
ones with a `.index.json` weight map) read by `WanModel.from_pretrained` (wan23/textimage2video.py:156) and, for the FSDP
sampler, copied parameter by parameter in fastvideo/utils/checkpoint.py:285-337. Here every tensor goes from the (lazily
mapped) file straight into the parameter that will hold it — on a 288 GB MI355X that parameter already lives in HBM, so the
host never holds more than one tensor and there is no second full copy of a 10-33 GB model.
"""
import json
import os

import torch

__all__ = ["weight_files", "stream_state_dict", "save_sharded"]


def weight_files(root, weights_name="diffusion_pytorch_model.safetensors"):
    """files holding the weights of a checkpoint directory, in load order."""
    idx = os.path.join(root, weights_name + ".index.json")
    if os.path.exists(idx):
        with open(idx) as fh:
            return sorted(set(json.load(fh)["weight_map"].values()))
    if os.path.exists(os.path.join(root, weights_name)):
        return [weights_name]
    for alt in (weights_name.replace(".safetensors", ".bin"), weights_name.replace(".safetensors", ".pth")):
        if os.path.exists(os.path.join(root, alt)):
            return [alt]
    raise FileNotFoundError(f"no {weights_name}[.index.json] (or .bin / .pth) under {root}")


def stream_state_dict(model, root, weights_name="diffusion_pytorch_model.safetensors"):
    """Copy every tensor of the checkpoint into the parameter / buffer of `model` with the same name (shape-checked; dtype and
    device follow the destination). Returns (missing_keys, unexpected_keys)."""
    targets = dict(model.named_parameters())
    targets.update(dict(model.named_buffers()))
    seen, unexpected = set(), []

    def put(name, src):
        dst = targets.get(name)
        if dst is None:
            unexpected.append(name)
            return
        if tuple(src.shape) != tuple(dst.shape):
            raise RuntimeError(f"{name}: checkpoint shape {tuple(src.shape)} != model shape {tuple(dst.shape)}")
        with torch.no_grad():
            dst.copy_(src)               # host -> HBM (+ dtype cast) in one pass
        seen.add(name)

    for f in weight_files(root, weights_name):
        path = os.path.join(root, f)
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(path, framework="pt", device="cpu") as sf:
                for name in sf.keys():
                    put(name, sf.get_tensor(name))
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True, mmap=True)
            for name in list(sd):
                put(name, sd.pop(name))
    return [k for k in targets if k not in seen], unexpected


def save_sharded(state_dict, root, weights_name="diffusion_pytorch_model.safetensors", max_shard_size=10 << 30):
    """safetensors shards of at most max_shard_size bytes (+ index) in the naming diffusers uses."""
    from safetensors.torch import save_file
    shards, cur, size = [], {}, 0
    for k, v in state_dict.items():
        n = v.numel() * v.element_size()
        if cur and size + n > max_shard_size:
            shards.append(cur)
            cur, size = {}, 0
        cur[k] = v.detach().contiguous()
        size += n
    if cur:
        shards.append(cur)
    if len(shards) == 1:
        save_file(shards[0], os.path.join(root, weights_name), metadata={"format": "pt"})
        return [weights_name]
    stem = weights_name[:-len(".safetensors")]
    names, weight_map, total = [], {}, 0
    for i, sh in enumerate(shards, 1):
        name = f"{stem}-{i:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(root, name), metadata={"format": "pt"})
        names.append(name)
        for k, v in sh.items():
            weight_map[k] = name
            total += v.numel() * v.element_size()
    with open(os.path.join(root, weights_name + ".index.json"), "w") as fh:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, fh, indent=2, sort_keys=True)
    return names
