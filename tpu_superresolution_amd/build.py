"""Build libsrk.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m tpu_superresolution_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are rebuilt only when their source (or a header) is
newer; the shared library lands next to this file so that it travels with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libsrk.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_stream.hip", "wgrad.hip", "convwgrad.hip", "attn.hip", "attn_fused.hip", "attn_bwd_fused.hip", "block_light.hip", "ln.hip", "misc.hip", "swinir.hip", "attn256.hip", "hat.hip", "hat_train.hip", "attn256_bwd.hip", "metrics.hip", "dat.hip", "dat_train.hip", "attn_rect_bwd.hip", "dat_small.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-result",
         "-Wno-subobject-linkage"]
FLAGS += os.environ.get("SRK_EXTRA_FLAGS", "").split()   # developer experiments (-DSRK_NT_GEMM=1 ...); touch the source to rebuild


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newest_header() -> float:
    t = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src: str, force: bool, hdr_time: float) -> str:
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    s = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(s), hdr_time):
        return obj
    cmd = [_hipcc(), *FLAGS, "-c", s, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(BUILD, exist_ok=True)
    hdr_time = _newest_header()
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_time), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
