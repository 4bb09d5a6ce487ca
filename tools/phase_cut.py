#!/usr/bin/env python3
"""Per-phase profile of the one-step kernel WITHOUT instrumenting it: the kernel is cut short at a phase boundary by patching
`s_endpgm` over the instruction that follows the boundary in a copy of the library, and the copies are run under rocprofv3 - the
difference between two cuts is what the phase in between executes (instructions, by the SQ counters) and costs (launch duration,
by the kernel trace).  No stamps, no waits, no extra stores: every instruction that runs is an instruction of the build.

The boundaries are the CZ_STAMP(i) positions of cz_kernels.h.  `make -C cooking_zoo_amd/csrc markers` builds the library with an
assembler COMMENT there (and the listing cz_inst_small_mark.s); comments are not instructions, but they are scheduling barriers,
so this "marker build" is the shipped code up to a handful of instructions (tools/isa_diff.py says how many: 3893 against 3895
static instructions for k_step<1,1,2,3,0> in round 5).

    python3 tools/phase_cut.py make [KERNEL_SUBSTRING]     # build container or GPU box: writes cooking_zoo_amd/csrc/cuts/libcz_cut_<i>.so
    bash tools/phase_cut.sh                                # GPU box: runs them (tools/phase_cut_run.py) under rocprofv3
    python3 tools/phase_cut.py table DIR                   # the table from DIR/cut_<i>/...
"""
import collections
import csv
import difflib
import glob
import json
import os
import re
import struct
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "cooking_zoo_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PHASES = {0: "entry: ids, addresses, first loads issued", 1: "prologue: record / table / descriptor loads, LDS table, barrier",
          2: "agents: orientation, targets, collisions, walking, interactions", 3: "progress_world + linked interactions",
          4: "recipe marks, rewards, termination / truncation, despawn / respawn", 5: "outputs: rewards / flags stored, statistics",
          6: "observe: LDS image + float64 rows", 7: "write-back of the record"}
S_ENDPGM = struct.pack("<I", 0xBF810000)


def bundles(blob):
    """(offset of the bundle, [(triple, offset, size)]) for every clang offload bundle in a file"""
    out = []
    for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__", blob):
        b = m.start()
        n = struct.unpack_from("<Q", blob, b + 24)[0]
        p = b + 32
        ents = []
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            ents.append((triple, off, size))
            p += 24 + tl
        out.append((b, ents))
    return out


def listing_positions(listing, mangled):
    """mnemonics of the kernel in the -S listing, and for every marker i the index of the first instruction behind it"""
    txt = open(listing).read()
    m = re.search(r"^" + re.escape(mangled) + r":[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, flags=re.S | re.M)
    mn, marks = [], {}
    for line in m.group(1).split("\n"):
        k = re.match(r"\s*; CZ_MARK (\d+)", line)
        if k:
            marks.setdefault(int(k.group(1)), len(mn))
        elif line.startswith("\t") and not line.strip().startswith((".", ";")):
            mn.append(line.split()[0])
    return mn, marks


def make(kernel_sub):
    lib = os.path.join(CSRC, "libcookingzoo_hip_mark.so")
    listing = os.path.join(CSRC, "cz_inst_small_mark.s")
    if not (os.path.exists(lib) and os.path.exists(listing)):
        subprocess.check_call(["make", "-C", CSRC, "markers"])
    blob = open(lib, "rb").read()
    names = re.findall(r"^(_ZN2cz6k_step\w+):", open(listing).read(), flags=re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True).stdout.split("\n")
    mangled = [n for n, d in zip(names, dem) if kernel_sub in d.replace(" ", "")]
    assert len(mangled) == 1, (kernel_sub, mangled)
    mangled = mangled[0]
    mn_s, marks = listing_positions(listing, mangled)
    # the code object that holds this kernel
    found = None
    for b, ents in bundles(blob):
        for triple, off, size in ents:
            if "gfx950" not in triple or size == 0:
                continue
            co = blob[b + off:b + off + size]
            tmp = "/tmp/cz_mark_%d.co" % (b + off)
            open(tmp, "wb").write(co)
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f"--disassemble-symbols={mangled}", tmp], capture_output=True, text=True).stdout
            if mangled in dis and len(dis.splitlines()) > 100:
                found = (b + off, tmp, dis)
    assert found, "kernel not found in any gfx950 code object of the marker library"
    base, tmp, dis = found
    ins = []                                                         # (address, mnemonic)
    for line in dis.splitlines():
        k = re.match(r"\s+(\S+)\s.*// ([0-9A-Fa-f]+):", line)
        if k:
            ins.append((int(k.group(2), 16), k.group(1)))
    # map listing positions to object addresses (the object has alignment padding the listing writes as directives)
    sm = difflib.SequenceMatcher(a=mn_s, b=[m for _, m in ins], autojunk=False)
    amap = {}
    for blk in sm.get_matching_blocks():
        for k in range(blk.size):
            amap[blk.a + k] = blk.b + k
    # .text: file offset of an address inside the code object
    sec = subprocess.run([OBJDUMP.replace("objdump", "readelf"), "-S", tmp], capture_output=True, text=True).stdout
    t = re.search(r"\]\s+\.text\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
    vma, off, size = int(t.group(1), 16), int(t.group(2), 16), int(t.group(3), 16)
    os.makedirs(os.path.join(CSRC, "cuts"), exist_ok=True)
    meta = {"kernel": kernel_sub, "static_instructions": len(mn_s), "cuts": {}}
    for i, pos in sorted(marks.items()):
        if i not in PHASES:
            continue
        while pos not in amap:
            pos += 1
        addr = ins[amap[pos]][0]
        fo = base + off + (addr - vma)
        assert vma <= addr < vma + size
        patched = bytearray(blob)
        patched[fo:fo + 4] = S_ENDPGM
        out = os.path.join(CSRC, "cuts", f"libcz_cut_{i}.so")
        open(out, "wb").write(patched)
        os.chmod(out, 0o755)
        meta["cuts"][i] = {"listing_index": marks[i], "address": hex(addr), "was": ins[amap[pos]][1]}
    json.dump(meta, open(os.path.join(CSRC, "cuts", "cuts.json"), "w"), indent=1)
    print(json.dumps(meta, indent=1))


def table(root):
    meta = json.load(open(os.path.join(root, "cuts.json"))) if os.path.exists(os.path.join(root, "cuts.json")) else {}
    rows = {}
    for d in sorted(glob.glob(os.path.join(root, "cut_*"))):
        i = os.path.basename(d)[4:]
        c = collections.defaultdict(list)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_step<" in r["Kernel_Name"]:
                    c[r["Counter_Name"]].append(float(r["Counter_Value"]) / (int(r["Grid_Size"]) // 64))
        dur = []
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_step<" in r["Kernel_Name"]:
                    dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        rows[i] = {k: sum(v) / len(v) for k, v in c.items()}
        if dur:
            dur = sorted(dur)[len(dur) // 10: len(dur) - len(dur) // 10] or dur
            rows[i]["launch_us"] = sum(dur) / len(dur) / 1e3
    order = [str(i) for i in sorted(PHASES)] + ["full"]
    keys = ["SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"]
    print(f"# Phase table of cz::{meta.get('kernel', '?')} (one launch per step, 4096 envs, bench workload, warm state) - `bash tools/phase_cut.sh`\n")
    print("Marker build (assembler comments at the CZ_STAMP boundaries; tools/isa_diff.py: a handful of instructions off the shipped kernel) cut short by\n"
          "`s_endpgm` at one boundary per library copy; rocprofv3 `--pmc` (two counter sets, 64 launches each) and `--kernel-trace` (2000 direct launches,\n"
          "trimmed mean) per copy.  Counts are exact and additive: per wave (= env) and launch, what ran up to the boundary, and the phase's own share as\n"
          "the difference to the row above.  Launch times are NOT additive - a kernel that ends early also ends the contention its later phases would have\n"
          "caused, and a launch ends with its slowest wave - they say how long a launch takes whose waves all stop there.\n")
    print("| kernel ends at | last phase that ran | SALU | VALU | LDS | SMEM | VMEM rd / wr | branch | launch us | this phase: SALU + VALU + LDS | + us |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    prev = None
    out = {}
    for i in order:
        if i not in rows or "SQ_INSTS_SALU" not in rows[i]:
            continue
        r = rows[i]
        g = lambda k: r.get(k, float("nan"))
        name = "whole kernel" if i == "full" else f"boundary {i}"
        last = PHASES[int(i)] if i != "full" else "(behind boundary 7: s_endpgm only)"
        p0 = prev or {}
        own = f"{g('SQ_INSTS_SALU') - p0.get('SQ_INSTS_SALU', 0):.0f} + {g('SQ_INSTS_VALU') - p0.get('SQ_INSTS_VALU', 0):.0f} + {g('SQ_INSTS_LDS') - p0.get('SQ_INSTS_LDS', 0):.0f}"
        own_us = f"{g('launch_us'):.2f}" if prev is None else f"{g('launch_us') - p0.get('launch_us', float('nan')):+.2f}"
        print(f"| {name} | {last} | " + " | ".join(f"{g(k):.0f}" for k in keys) + f" | {g('SQ_INSTS_VMEM_RD'):.0f} / {g('SQ_INSTS_VMEM_WR'):.0f} | "
              f"{g('SQ_INSTS_BRANCH'):.0f} | {g('launch_us'):.2f} | {own} | {own_us} |")
        out[i] = r
        prev = r
    json.dump(out, open(os.path.join(root, "phase_table.json"), "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "make":
        make(sys.argv[2] if len(sys.argv) > 2 else "k_step<1,1,2,3,0>")
    else:
        table(sys.argv[2])
