"""Register / spill / scratch report of the gfx950 kernels inside a built object (nu_nerf_amd/build/*.o).

    python3 scripts/kernel_regs.py [gemm_nt16 ...]        (default: every object of the last `python -m nu_nerf_amd.build`)

Reads the code-object metadata (llvm-readelf --notes) of the device ELF embedded in the host object's .hip_fatbin section;
tests/test_abi.py::test_default_kernels_do_not_spill uses `kernel_table` to hold the hot kernels to zero VGPR spills."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table(obj_path):
    """[(demangled kernel name, {vgpr, vgpr_spill, sgpr, sgpr_spill, scratch, lds})] of one host object."""
    with tempfile.TemporaryDirectory() as td:
        fat, elf = os.path.join(td, "fatbin"), os.path.join(td, "dev.elf")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], text=True)
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        def f(key):
            return int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out.append((name, dict(vgpr=f("vgpr_count"), vgpr_spill=f("vgpr_spill_count"), sgpr=f("sgpr_count"),
                               sgpr_spill=f("sgpr_spill_count"), scratch=f("private_segment_fixed_size"),
                               lds=f("group_segment_fixed_size"))))
    names = subprocess.check_output(["c++filt"], input="\n".join(n for n, _ in out), text=True).split("\n")
    return [(re.sub(r"^void ", "", d), r) for d, (_, r) in zip(names, out)]


def instruction_count(obj_path, pattern):
    """Number of disassembled device instructions of one host object whose mnemonic matches the regular expression `pattern`."""
    with tempfile.TemporaryDirectory() as td:
        fat, elf = os.path.join(td, "fatbin"), os.path.join(td, "dev.elf")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat])
        rc = subprocess.call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                              "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf], stderr=subprocess.DEVNULL)
        if rc != 0:          # a host-only object (csrc/mlp_ops.hip sequences launches, it holds no kernel)
            return 0
        dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", elf], text=True)
    rx = re.compile(r"^\s+(" + pattern + r")\b")
    return sum(1 for line in dis.splitlines() if rx.match(line))


if __name__ == "__main__":
    bdir = os.path.join(ROOT, "nu_nerf_amd", "build")
    objs = sys.argv[1:] or sorted(f[:-2] for f in os.listdir(bdir) if f.endswith(".o"))
    for o in objs:
        for name, r in kernel_table(os.path.join(bdir, o + ".o")):
            print(f"{o:10s} {name[:64]:64s} vgpr {r['vgpr']:3d} spill {r['vgpr_spill']:3d}  sgpr {r['sgpr']:3d} spill {r['sgpr_spill']:3d}"
                  f"  scratch {r['scratch']:4d} B  lds {r['lds']:6d} B")
