"""Build libmmf_amd.so (gfx950 device code + C ABI) in-tree with hipcc.

    python -m mmf_amd.csrc.build            # incremental
    python -m mmf_amd.csrc.build --force
    MMF_AMD_EXTRA_HIPCC_FLAGS=-DMMF_ATTN_PROBE python -m mmf_amd.csrc.build --tag probe
                                            # instrumented copy: objects *.probe.o, mmf_amd/libmmf_amd.probe.so (load it with
                                            # MMF_AMD_LIB=...); the regular library is left alone

hipcc cross-compiles for gfx950 without a GPU; the resulting .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
SOURCES = ["lib.hip", "gemm.hip", "attention.hip", "rowops.hip", "m4c_ops.hip", "gate_ops.hip", "transpose.hip", "fp32_path.hip", "fp32_train.hip", "uniter_ops.hip"]
LIB = os.path.join(os.path.dirname(HERE), "libmmf_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
    "-Wno-unused-result", "-I", INCLUDE, "-I", HERE,
] + os.environ.get("MMF_AMD_EXTRA_HIPCC_FLAGS", "").split()     # e.g. -DMMF_WIDE_ABLATE for tools/wide_ablate.py


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src, os.path.join(INCLUDE, "mmf_amd.h"), __file__] + [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force, tag=""):
    obj = os.path.join(HERE, src.replace(".hip", tag + ".o"))
    path = os.path.join(HERE, src)
    if force or _stale(obj, path):
        cmd = [HIPCC, *FLAGS, "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True, tag=""):
    tag = "." + tag if tag else ""
    LIB = os.path.join(os.path.dirname(HERE), "libmmf_amd%s.so" % tag)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, tag), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    if not tag:
        build_ops(force=force, verbose=verbose)
    return LIB


OPS_SRC = os.path.join(HERE, "torch_ops.cpp")
OPS_LIB = os.path.join(os.path.dirname(HERE), "libmmf_amd_ops.so")


def build_ops(force=False, verbose=True):
    """libmmf_amd_ops.so: the native PyTorch operator library (torch_ops.cpp — TORCH_LIBRARY(mmf_amd) + C++ autograd nodes over the C ABI).
    Host code only (g++ against the installed torch headers; the device code is all in libmmf_amd.so, found through $ORIGIN)."""
    import torch
    deps = [OPS_SRC, os.path.join(INCLUDE, "mmf_amd.h"), __file__]
    if not force and os.path.exists(OPS_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(OPS_LIB) for d in deps):
        if verbose:
            print("built", OPS_LIB)
        return OPS_LIB
    ti = os.path.dirname(torch.__file__)
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi, "-I", os.path.join(ti, "include"), "-I", os.path.join(ti, "include", "torch", "csrc", "api", "include"),
           "-I", "/opt/rocm/include", "-I", INCLUDE, OPS_SRC, "-o", OPS_LIB, "-L", os.path.dirname(HERE), "-lmmf_amd",
           "-L", os.path.join(ti, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-Wl,-rpath,$ORIGIN",
           "-Wl,-rpath," + os.path.join(ti, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building %s failed:\n%s\n%s" % (OPS_LIB, r.stdout, r.stderr))
    if verbose:
        print("built", OPS_LIB)
    return OPS_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, tag=sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "")
