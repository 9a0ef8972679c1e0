"""TEST INFRASTRUCTURE ONLY — imports the REAL reference modules from /root/reference.

Only usable in the build container (the GPU box has no /root/reference). Used by
oracle/make_golden.py to generate tests/golden/* and by the `-m "not gpu"` tests that pin the
oracle restatement (oracle/dit.py, oracle/vae.py) against the reference itself.

Recipe: stub the 4 diffusers symbols the model files import, import
wan{,23}/modules/{attention,model}.py by file path under a synthetic package (the real package
__init__ pulls torchvision/easydict/peft which are not installed), and rebind `flash_attention`
(needs CUDA + the un-vendored flash-attn 2.7.0.post2 wheel) to an exact-softmax restatement that
follows wan/modules/attention.py:56-130.
"""
import importlib.util
import math
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("YUME_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "wan23", "modules", "model.py"))


def _stub_diffusers():
    if "diffusers" in sys.modules and not getattr(sys.modules["diffusers"], "_yume_stub", False):
        return
    d = types.ModuleType("diffusers")
    d._yume_stub = True
    cu = types.ModuleType("diffusers.configuration_utils")
    cu.ConfigMixin = type("ConfigMixin", (), {})
    cu.register_to_config = lambda f: f
    dm = types.ModuleType("diffusers.models")
    mu = types.ModuleType("diffusers.models.modeling_utils")
    mu.ModelMixin = type("ModelMixin", (nn.Module,), {})
    sys.modules.update({"diffusers": d, "diffusers.configuration_utils": cu, "diffusers.models": dm,
                        "diffusers.models.modeling_utils": mu})


def sdpa_standin(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
                 window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, version=None):
    """Exact softmax attention with flash_attention()'s signature (attention.py:24-38): q,k,v [B,L,N,D],
    scale 1/sqrt(D), k_lens honoured by truncation, output in q.dtype. fp32 gold: no bf16 cast."""
    assert not causal and dropout_p == 0. and q_lens is None
    out_dtype = q.dtype
    b = q.size(0)
    outs = []
    for i in range(b):
        lk = int(k_lens[i]) if k_lens is not None else k.size(1)
        qi = q[i].transpose(0, 1).double()
        ki = k[i, :lk].transpose(0, 1).double()
        vi = v[i, :lk].transpose(0, 1).double()
        if q_scale is not None:
            qi = qi * q_scale
        sc = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(q.size(-1))
        a = torch.softmax(qi @ ki.transpose(1, 2) * sc, dim=-1)
        outs.append((a @ vi).transpose(0, 1))
    return torch.stack(outs).to(out_dtype)


def _import_pkg(pkg, files):
    root = types.ModuleType(pkg)
    root.__path__ = [os.path.join(REF_ROOT, pkg)]
    sub = types.ModuleType(pkg + ".modules")
    sub.__path__ = [os.path.join(REF_ROOT, pkg, "modules")]
    sys.modules[pkg] = root
    sys.modules[pkg + ".modules"] = sub
    mods = {}
    for f in files:
        name = f"{pkg}.modules.{f}"
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, pkg, "modules", f + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        mods[f] = m
    return mods


_cache = {}


def ref_dit(family="wan23"):
    """Returns the reference `model` module of wan23 (5B arch) or wan (14B arch), flash_attention rebound."""
    assert available(), "reference tree not present"
    key = ("dit", family)
    if key not in _cache:
        _stub_diffusers()
        mods = _import_pkg(family, ["attention", "model"])
        mods["model"].flash_attention = sdpa_standin
        _cache[key] = mods["model"]
    return _cache[key]


def ref_vae(which="vae2_2"):
    """Returns the reference VAE module: 'vae2_2' (wan23/modules/vae2_2.py) or 'vae2_1' (wan/modules/vae.py)."""
    assert available(), "reference tree not present"
    key = ("vae", which)
    if key not in _cache:
        path = {"vae2_2": os.path.join(REF_ROOT, "wan23", "modules", "vae2_2.py"),
                "vae2_1": os.path.join(REF_ROOT, "wan", "modules", "vae.py")}[which]
        spec = importlib.util.spec_from_file_location("yume_ref_" + which, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        _cache[key] = m
    return _cache[key]


def ref_t5():
    """Returns the reference text-encoder module wan/modules/t5.py (its `.tokenizers` import — HuggingFace tokenizer glue with
    ftfy/regex dependencies — is replaced by an empty stand-in: only the nn.Modules are used)."""
    assert available(), "reference tree not present"
    if "t5" not in _cache:
        pkg = "yume_ref_t5pkg"
        root = types.ModuleType(pkg)
        root.__path__ = [os.path.join(REF_ROOT, "wan", "modules")]
        tok = types.ModuleType(pkg + ".tokenizers")
        tok.HuggingfaceTokenizer = type("HuggingfaceTokenizer", (), {})
        sys.modules[pkg] = root
        sys.modules[pkg + ".tokenizers"] = tok
        spec = importlib.util.spec_from_file_location(pkg + ".t5", os.path.join(REF_ROOT, "wan", "modules", "t5.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[pkg + ".t5"] = m
        # t5.py:478 evaluates torch.cuda.current_device() as a default argument at import time: no GPU in this container
        real = torch.cuda.current_device
        torch.cuda.current_device = lambda: 0
        try:
            spec.loader.exec_module(m)
        finally:
            torch.cuda.current_device = real
        _cache["t5"] = m
    return _cache["t5"]


def ref_clip():
    """Returns the reference wan/modules/clip.py (vision tower). Stand-ins: torchvision.transforms (not installed; only used to
    build the PIL preprocessing pipeline), `.tokenizers`, `.xlm_roberta` (text tower base class, never instantiated here) and
    `.attention.flash_attention` (exact softmax, as for the DiT)."""
    assert available(), "reference tree not present"
    if "clip" not in _cache:
        if "torchvision" not in sys.modules:
            tv = types.ModuleType("torchvision")
            tvt = types.ModuleType("torchvision.transforms")
            tv.transforms = tvt
            sys.modules["torchvision"] = tv
            sys.modules["torchvision.transforms"] = tvt
        pkg = "yume_ref_clippkg"
        root = types.ModuleType(pkg)
        root.__path__ = [os.path.join(REF_ROOT, "wan", "modules")]
        tok = types.ModuleType(pkg + ".tokenizers")
        tok.HuggingfaceTokenizer = type("HuggingfaceTokenizer", (), {})
        xlm = types.ModuleType(pkg + ".xlm_roberta")
        xlm.XLMRoberta = type("XLMRoberta", (nn.Module,), {})
        att = types.ModuleType(pkg + ".attention")
        att.flash_attention = sdpa_standin
        sys.modules.update({pkg: root, pkg + ".tokenizers": tok, pkg + ".xlm_roberta": xlm, pkg + ".attention": att})
        spec = importlib.util.spec_from_file_location(pkg + ".clip", os.path.join(REF_ROOT, "wan", "modules", "clip.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[pkg + ".clip"] = m
        spec.loader.exec_module(m)
        _cache["clip"] = m
    return _cache["clip"]


class _DistWithListAllToAll:
    """torch.distributed for the reference's wan23/distributed/util.py on the gloo backend (CPU tests): gloo has no list all_to_all, so
    that ONE call is served by all_to_all_single (same collective: outputs[i] <- what rank i put in its inputs[my rank]); everything else
    is torch.distributed's."""

    def __getattr__(self, name):
        import torch.distributed as dist
        return getattr(dist, name)

    @staticmethod
    def all_to_all(outputs, inputs, group=None, **kw):
        import torch.distributed as dist
        send = torch.stack([u.contiguous() for u in inputs])
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv.view(torch.uint8), send.view(torch.uint8), group=group)
        for o, r in zip(outputs, recv):
            o.copy_(r)


def ref_ulysses():
    """the reference's wan23/distributed/ulysses.py (`distributed_attention`) with its flash_attention bound to the exact-softmax stand-in
    and its util's `dist` to the shim above. Needs an initialised process group when called."""
    assert available(), "reference tree not present"
    key = ("ulysses",)
    if key not in _cache:
        _stub_diffusers()
        _import_pkg("wan23", ["attention"])
        sub = types.ModuleType("wan23.distributed")
        sub.__path__ = [os.path.join(REF_ROOT, "wan23", "distributed")]
        sys.modules["wan23.distributed"] = sub
        mods = {}
        for f in ("util", "ulysses"):
            name = "wan23.distributed." + f
            spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, "wan23", "distributed", f + ".py"))
            m = importlib.util.module_from_spec(spec)
            sys.modules[name] = m
            spec.loader.exec_module(m)
            mods[f] = m
        mods["util"].dist = _DistWithListAllToAll()
        mods["ulysses"].flash_attention = sdpa_standin
        _cache[key] = mods["ulysses"]
    return _cache[key]
