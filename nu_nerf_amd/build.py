"""Build the gfx950 HIP library in-tree: nu_nerf_amd/libnunerf.so.

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box
with the gpurun snapshot.  The library has no torch dependency: its C ABI (include/nu_nerf.h) takes
raw device pointers + a hipStream_t.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnunerf.so")
STAMP = os.path.join(HERE, ".libnunerf.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-fno-gpu-rdc", "-Wno-unused-result", "-I" + CSRC, "-I" + os.path.join(HERE, "..", "include")]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode())
                h.update(fh.read())
    inc = os.path.join(HERE, "..", "include", "nu_nerf.h")
    if os.path.exists(inc):
        with open(inc, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB
    objs = []
    procs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
