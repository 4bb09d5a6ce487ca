#!/usr/bin/env python3
"""Compare two `hipcc -S --cuda-device-only` listings kernel by kernel: instruction counts, and whether the instruction
streams are identical (labels and comments ignored).  Used when a refactoring is meant to leave the shipped kernels alone.

Usage: python tools/isa_diff.py before.s after.s [--show NAME_SUBSTRING]"""
import re
import subprocess
import sys


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, flags=re.S | re.M):
        ins = [re.sub(r"\s*;.*$", "", l.strip()) for l in m.group(2).split("\n")
               if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        out[m.group(1)] = ins
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    names = sorted(set(a) | set(b))
    dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True).stdout.split("\n")))
    short = lambda n: re.sub(r"\(.*", "", dem[n]).replace("void ", "")
    differing = 0
    for n in names:
        if n not in a or n not in b:
            print(f"{'only before' if n in a else 'only after ':12s} {short(n):40s} {len(a.get(n, b.get(n)))}")
        elif a[n] != b[n]:
            differing += 1
            print(f"{'DIFFERENT':12s} {short(n):40s} {len(a[n])} -> {len(b[n])}")
    print(f"{len(a)} kernels before, {len(b)} after, {differing} with a different instruction stream")


if __name__ == "__main__":
    main()
