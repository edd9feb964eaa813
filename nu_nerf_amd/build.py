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
# NU_BUILD_NO_PK_F32=1: build WITHOUT packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 ...).  A pure-VALU kernel of this library
# built WITH them returned wrong elements (lanes 48-63 of a wave) in 59 launches of 60 while the library's bf16-MFMA GEMMs ran on another
# stream, and never without them (scripts/determinism_valu_victim.py, profiles/r04/valu_victim_next_to_bf16_mfma.txt); the two
# builds are equally fast.  It is NOT the default: without the packed multiplies the compiler contracts other expressions (o + d z of the
# sample positions becomes an fma; one ulp there is 3e-5 in sin(512 x) of the NeRF++ embedding) and three reference-parity tests land
# 4-6e-4 from the golden values on one weight gradient, past the 3e-4 they were calibrated to with the packed build.  The default build
# avoids the failing combination instead: the bf16 / bf16x6 modes never overlap two streams (engine.render_forward, stage2.py), and the
# fp32-MFMA kernels of the default mode do not trigger it (0 of 60).
if os.environ.get("NU_BUILD_NO_PK_F32", "0") == "1":
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


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
