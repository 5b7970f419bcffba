"""Build libvct_hip.so (all HIP kernels, gfx950) in-tree with hipcc.  No JIT cache, no torch
extension machinery: the product boundary is a plain C-ABI shared library (include/vct_hip.h)."""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libvct_hip.so")
SOURCES = ["vct_gemm.hip", "vct_gemm_bf16.hip", "vct_gemm_bf16_nt.hip", "vct_gemm_bf16_nn.hip", "vct_gemm_bf16_tn.hip", "vct_gemm256.hip", "vct_gemm_pt.hip", "vct_gemm_skinny.hip", "vct_attn.hip", "vct_layer_ss.hip", "vct_layer_ss_bwd.hip", "vct_norm.hip", "vct_elem.hip", "vct_optim.hip", "vct_decode.hip", "vct_decode_block.hip", "vct_runtime.hip", "vct_comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _read(path):
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return f.read()


def _digest():
    h = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)) + ["../../include/vct_hip.h"]:
        p = os.path.join(CSRC, fn)
        if os.path.isfile(p):
            h.update(fn.encode())
            with open(p, "rb") as f:
                h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip for gfx950 and link libvct_hip.so next to this file."""
    stamp = os.path.join(PKG, "build", "stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and _read(stamp) == dig:
        return LIB
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    # objects (and their stamps) of sources that are no longer part of the library: unlinked, but build/ ships to the GPU box
    keep = {s.replace(".hip", ".o") for s in srcs}
    for fn in os.listdir(os.path.join(PKG, "build")):
        base = fn[:-len(".stamp")] if fn.endswith(".stamp") else fn
        if base.endswith(".o") and base not in keep:
            os.remove(os.path.join(PKG, "build", fn))

    # per-object stamps: a source is recompiled when it, any header, or the flags changed (editing one .hip rebuilds one object)
    hh = hashlib.sha256(" ".join(FLAGS).encode())
    for fn in sorted(os.listdir(CSRC)) + ["../../include/vct_hip.h"]:
        if fn.endswith(".h"):
            with open(os.path.join(CSRC, fn), "rb") as f:
                hh.update(fn.encode() + f.read())
    hdr_dig = hh.hexdigest()

    def cc(src):
        obj = os.path.join(PKG, "build", src.replace(".hip", ".o"))
        with open(os.path.join(CSRC, src), "rb") as f:
            odig = hashlib.sha256(hdr_dig.encode() + f.read()).hexdigest()
        if not force and os.path.exists(obj) and _read(obj + ".stamp") == odig:
            return obj
        cmd = [_hipcc(), *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        with open(obj + ".stamp", "w") as f:
            f.write(odig)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
