#!/usr/bin/env python3
"""Register / LDS / scratch metadata of every kernel in the shipped library.

Compiles each kernel unit to gfx950 assembly with the Makefile's flags (`hipcc -S --cuda-device-only`; no GPU
needed) and prints the `.amdgpu_metadata` fields per kernel plus the static instruction count of its body.

    python3 tools/kernel_metadata.py > profiles/r03/kernel_metadata.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cooking_zoo_amd", "csrc")
UNITS = ["cz_inst_small.hip", "cz_inst_large.hip", "cz_inst_huge.hip", "cz_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
         "-mllvm", "-amdgpu-kernarg-preload-count=14"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                         capture_output=True, check=True).stdout.split("\n")
    return dict(zip(names, out))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("cz::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\s+", "", name)


def body_sizes(asm):
    """static instruction count per kernel symbol: lines between `sym:` and its `s_endpgm` that look like code"""
    sizes, cur, n = {}, None, 0
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, n = m.group(1), 0
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        n += 1
        if s.startswith("s_endpgm"):
            sizes.setdefault(cur, n)
            cur = None
    return sizes


def main():
    extra = sys.argv[1:]
    print("Register / LDS / scratch metadata of every kernel in the shipped library (hipcc " + " ".join(FLAGS[:2]) +
          ", ROCm 7.2;\nfields from the .amdgpu_metadata notes; `instr` = static instructions up to the first s_endpgm).  "
          "k_step<OPL,CPL,NA,SCHEME,FUSED>;\nk_step_chain<OPL,CPL,NA,SCHEME> exists for the small instance with up to three agents only and is "
          "compiled for eight waves per SIMD.\n")
    with tempfile.TemporaryDirectory() as tmp:
        for unit in UNITS:
            out = os.path.join(tmp, unit + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-o", out,
                            os.path.join(CSRC, unit)], check=True, capture_output=True)
            asm = open(out).read()
            sizes = body_sizes(asm)
            kernels = []
            for blk in re.split(r"\n\s*- \.agpr_count:", asm)[1:]:
                f = {}
                for key in ("name", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count",
                            "sgpr_spill_count", "vgpr_count", "vgpr_spill_count"):
                    m = re.search(r"\.%s:\s*(\S+)" % key, blk)
                    f[key] = m.group(1) if m else "?"
                kernels.append(f)
            names = demangle([k["name"] for k in kernels])
            print("== %s" % unit)
            print("%-44s %6s %7s %5s %8s %5s %8s %6s" % ("kernel", "LDS", "scratch", "SGPR", "s.spill", "VGPR", "v.spill",
                                                          "instr"))
            for k in kernels:
                print("%-44s %6s %7s %5s %8s %5s %8s %6s" % (
                    short(names[k["name"]])[:44], k["group_segment_fixed_size"], k["private_segment_fixed_size"],
                    k["sgpr_count"], k["sgpr_spill_count"], k["vgpr_count"], k["vgpr_spill_count"],
                    sizes.get(k["name"], "?")))
            print()


if __name__ == "__main__":
    main()
