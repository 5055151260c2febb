"""Builds libopenmatch_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

No torch involvement: the library is plain HIP + a C ABI (include/openmatch_hip.h); the
Python side binds it with ctypes.  `python -m openmatch_amd._build` rebuilds it.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
REPO = os.path.dirname(HERE)
OBJ_DIR = os.path.join(REPO, "build", "obj")
LIB_PATH = os.path.join(CSRC, "libopenmatch_hip.so")
ARCH = "gfx950"
SOURCES = ["abi.cpp", "comm.cpp", "gemm.hip", "gemm_wide6_bf16.hip", "gemm_wide6_f32.hip", "gemm_wide7.hip", "gemm_wide7_ln.hip", "gemm_wide7_f16.hip", "gemm_tn.hip", "elementwise.hip", "attention.hip", "attention_bwd16.hip", "encoder.hip", "decoder.hip",
           "search.hip", "contrastive.hip", "train_kernels.hip", "train.hip", "optim.hip", "gemm_skinny.hip"]
HEADERS = ["common.h", "gemm_core.h", "gemm_core2.h", "gemm_core6.h", "gemm_core7.h", "gemm_wide7.h", "gemm_wide6.h", "gemm_epilogue.h", "gemm_epilogue6.h", "attn_common.h", "train_kernels.h", "kernels.h", os.path.join("..", "..", "include", "openmatch_hip.h")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X build needs ROCm's hipcc on PATH")
    return exe


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths if os.path.exists(p))


def build_native(force=False, verbose=False, probe=False):
    """Compile every translation unit for gfx950 and link the shared library. Returns its path.
    probe=True (`--probe`): also compile the timing variants of the attention / small-batch scan kernels that skip part of their
    work (-DOM_PROBE_KERNELS; selectable with OM_ATTENTION_DEBUG / OM_SEARCH_DEBUG bits 3-4, results WRONG) -- never shipped."""
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-x", "hip", "-Wno-unused-result"]
    if probe:
        flags.append("-DOM_PROBE_KERNELS")
        force = True
    hdr_time = _newest(hdrs)

    def compile_one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time)):
            return obj, False
        cmd = [hipcc] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj, True

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(compile_one, srcs))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True, probe="--probe" in sys.argv))
