#!/usr/bin/env python3
"""Build-time check of the one place where k_render issues loads hipcc does not know of (LAYOUT_POINT_KEYS, PCR_PK_LOAD_ROW /
PCR_PK_WAIT in pcr_kernels.hip.h): between the three `buffer_load_ushort` written as inline assembly and the `s_waitcnt vmcnt(0)`
that stands in front of the first use of their results, no instruction may touch the three destination registers -- a copy, a
spill or a reuse would read or clobber a register a load is still in flight into.

    python tools/check_keys_asm.py [file.s]      (without a file: compiles pcr_api.hip with --save-temps into a temporary directory)

Checks every k_render<*, LAYOUT_POINT_KEYS, false, *> of the listing (the checked variant keeps compiler-known loads): from the first inline-assembly load to the last inline-assembly
wait of the function, a line that names one of the destination registers must be one of the loads, or stand between a wait and the
next load block (the table reads that use the keys up, the checked variant's copies of them)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def listing(path=None):
    if path:
        return open(path).read()
    d = tempfile.mkdtemp(prefix="pcr_keys_asm_")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "pcrhpg24_amd", "csrc"), "--save-temps", "-c", os.path.join(ROOT, "pcrhpg24_amd", "csrc", "pcr_api.hip"),
                    "-o", os.path.join(d, "api.o")], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(os.path.join(d, "pcr_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def names(line, regs):
    """does the line name one of the vector registers (alone or inside a range v[a:b])?"""
    for m in re.finditer(r"\bv(\d+)\b", line):
        if int(m.group(1)) in regs:
            return True
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        if any(int(m.group(1)) <= r <= int(m.group(2)) for r in regs):
            return True
    return False


def check(text):
    bad, seen = [], 0
    for m in re.finditer(r"^(_ZN3pcr8k_renderILi(\d)ELi2ELb(0)ELi(\d)EEEvNS_10RenderArgsE):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        seen += 1
        name = "k_render<%s, keys, %s, %s>" % (m.group(2), m.group(3), m.group(4))
        lines = m.group(5).split("\n")
        in_asm, kind = False, {}
        for i, l in enumerate(lines):
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif in_asm and "buffer_load_ushort" in l:
                kind[i] = "load"
            elif in_asm and "s_waitcnt vmcnt(0)" in l:
                kind[i] = "wait"
        loads = [i for i, k in kind.items() if k == "load"]
        waits = [i for i, k in kind.items() if k == "wait"]
        if len(loads) != 3 or not waits:
            bad.append("%s: expected one block of three loads and at least one wait, found %d loads, %d waits" % (name, len(loads), len(waits)))
            continue
        regs = set(int(re.search(r"buffer_load_ushort v(\d+),", lines[i]).group(1)) for i in loads)
        if len(regs) != 3:
            bad.append("%s: the three loads do not have three destinations" % name)
            continue
        first, last = min(loads), max(waits)
        # the part of the function the loads can be in flight in: from the top of the loop that holds them (the nearest loop header
        # above the first wait, or the first wait itself) to the last wait
        start = min(min(waits), first)
        for i in range(start, -1, -1):
            if "Loop Header" in lines[i]:
                start = i
                break
        safe = False                       # between a wait and the next block of loads
        for i in range(start, last + 1):
            if kind.get(i) == "wait":
                safe = True
            elif kind.get(i) == "load":
                safe = False
                continue
            l = lines[i].split(";")[0]
            if not safe and names(l, regs):
                bad.append("%s: line %d touches a key register while its load may be in flight: %s" % (name, i, lines[i].strip()))
    if seen == 0:
        bad.append("no k_render<*, LAYOUT_POINT_KEYS, false, *> in the listing")
    return seen, bad


if __name__ == "__main__":
    seen, bad = check(listing(sys.argv[1] if len(sys.argv) > 1 else None))
    for b in bad:
        print(b)
    print("%d kernels checked, %d findings" % (seen, len(bad)))
    sys.exit(1 if bad else 0)
