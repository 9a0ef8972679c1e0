"""Builds libyume_hip.so (gfx950) in-tree with hipcc. No cmake, no JIT cache: the .so travels with the tree.

    python -m yume_amd.build [--force]
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libyume_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=fast",
         "-Wno-unused-result", "-DNDEBUG"]
# per-file additions. attn_fwd7.hip hand-places one softmax piece per MFMA gap: the SLP vectoriser would fuse neighbouring f32 adds /
# multiplies into v_pk_* instructions, which cost more issue time beside MFMAs than the two scalar ones (MI355X_MICROARCH guide)
EXTRA_FLAGS = {"attn_fwd7.hip": ["-fno-slp-vectorize"], "attn_fwd8.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


LINK_FLAGS = ["-Wl,-Bsymbolic"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _fingerprint():
    h = hashlib.sha256()
    files = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "yume_hip.h")])
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())   # content-addressed: the tree may live at another path (GPU box)
            with open(f, "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    h.update(" ".join(LINK_FLAGS).encode())
    return h.hexdigest()


def build_library(force=False, verbose=True):
    """Compile every csrc/*.hip for gfx950 and link lib/libyume_hip.so. Returns the library path."""
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.stamp")
    fp = _fingerprint()
    if not force and os.path.exists(LIBPATH) and os.path.exists(stamp) and open(stamp).read().strip() == fp:
        return LIBPATH
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        # -save-temps=obj: the device assembly (obj/<name>-hip-amdgcn-amd-amdhsa-gfx950.s) stays next to the object; the ISA audits of the
        # hand-scheduled kernels (tests/test_attn7_isa.py, tests/test_gemm_w4_isa.py) read it instead of compiling the file again
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-I", INCLUDE, "-save-temps=obj", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        stem = os.path.join(objdir, os.path.basename(src)[:-4])
        for f in os.listdir(objdir):          # of the temporaries only the device assembly is kept
            full = os.path.join(objdir, f)
            if full.startswith(stem + "-") or full.startswith(stem + ".hip-"):
                if not f.endswith(f"-hip-amdgcn-amd-amdhsa-{ARCH}.s"):
                    os.remove(full)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    # -Bsymbolic: calls between the library's own translation units bind inside it (an experiment build of the library loaded next to
    # the product one — tools/*_check --lib — must run its own kernels, not the first-loaded library's)
    cmd = [hipcc, "-shared", "-fPIC"] + LINK_FLAGS + [f"--offload-arch={ARCH}", "-o", LIBPATH] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(fp)
    if verbose:
        print(f"[yume_amd.build] built {LIBPATH}")
    return LIBPATH


def device_asm(src_name):
    """Text of the gfx950 assembly hipcc produced for csrc/<src_name> in the current build (built first if it is stale)."""
    build_library(verbose=False)
    path = os.path.join(LIBDIR, "obj", src_name[:-4] + f"-hip-amdgcn-amd-amdhsa-{ARCH}.s")
    if not os.path.exists(path):      # an object directory from a build without -save-temps
        build_library(force=True, verbose=False)
    return open(path).read()


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
