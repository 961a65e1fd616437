#!/usr/bin/env python3
"""Per-opcode census of the point loop of a k_render instantiation, from the gfx950 assembly hipcc writes with --save-temps.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -I include -I pcrhpg24_amd/csrc --save-temps \
          pcrhpg24_amd/csrc/pcr_api.hip -o /tmp/isa/lib.so          (in an empty directory: the .s lands in the cwd)
    python tools/isa_census.py /tmp/isa/pcr_api-hip-amdgcn-amd-amdhsa-gfx950.s [mangled kernel name] > profiles/rNN_isa_census.md

The loop is the innermost loop of the kernel (the basic blocks the assembler comments tag with the deepest loop header).
Blocks are classed by what they hold: the IEEE division fallback, the global-memory scatter / pre-read of off-window points and
the double-precision dequantisation are side paths a wave of the benchmark frame does not execute (or skips with exec = 0);
everything else is the common path. Issue classes are the measured ones of tools/exp/instr_rate2.hip
(profiles/r02_instr_rate2.txt): `fast` 2.3 cycles per wave64 instruction per SIMD, `slow` 4.2, `v_rcp_f32` 8.2.
"""
import collections
import re
import sys

KERNEL = "_ZN3pcr8k_renderILi0ELi1ELb0EEEvNS_10RenderArgsE"
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32"}
CYCLES = {"fast": 2.3, "slow": 4.2, "rcp": 8.2}


def issue_class(op, operands):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"):
        return "rcp"
    sgpr = re.search(r"(^|[ ,\[|-])s(\d+|\[\d+:\d+\])", operands.split(",", 1)[1] if "," in operands else "") is not None or "vcc" in operands.split(",", 1)[-1]
    if base in FAST and not sgpr and not op.endswith("_sdwa") and not op.endswith("_dpp"):
        return "fast"
    return "slow"


def side_path(lines):
    text = "\n".join(lines)
    if "v_div_scale_f32" in text:
        return "IEEE division fallback (w outside [2^-64, 2^64))"
    if "global_atomic" in text:
        return "off-window scatter (global atomic)"
    if "global_load_dwordx2" in text and "v_mad_u64_u32" in text:
        return "off-window pre-read (global load)"
    if "global_store_byte" in text:
        return "tile of a stray point marked (dirty tiles)"
    if "global_load_dword" in text and "offset:4" in text:
        return "off-window pre-read (global load of the depth half; mostly_outside batches only)"
    if "v_subrev_u32" in text and "v_cmp_le_u32" in text and "ds_" not in text:
        return "is an off-window point a stray? (uniform branch: some lane is off its window)"
    if "v_cvt_f64_i32" in text:
        return "double-precision dequantisation (batches >= 100 px on screen)"
    return None


def main():
    path = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else KERNEL
    src = open(path).read().splitlines()
    start = next(i for i, l in enumerate(src) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(src)) if ".end_amdhsa_kernel" in src[i])
    body = src[start:end]
    # the innermost loop: the header named by the most "in Loop: Header=X Depth=N" comments of the greatest depth
    depth = collections.Counter()
    for l in body:
        m = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", l)
        if m:
            depth[(int(m.group(2)), m.group(1))] += 1
    top = max(d for d, _ in depth)
    header = max((n, h) for (d, h), n in depth.items() if d == top)[1]
    # split into basic blocks; keep those that belong to the loop
    blocks, cur, name, in_loop = [], [], None, False
    def flush():
        if cur and in_loop:
            blocks.append((name, list(cur)))
    for i, l in enumerate(body):
        m = re.match(r"^(\.L(BB\d+_\d+)):", l)
        m2 = re.match(r"^; %bb\.(\d+):", l)
        if m or m2:
            flush()
            cur, name = [], (m.group(2) if m else "bb." + m2.group(1))
            tag = " ".join(body[i:i + 3])
            in_loop = ("Header=" + header + " Depth=%d" % top) in tag or (m and m.group(2) == header)
            continue
        if re.match(r"^\s+[a-z]", l) and not l.strip().startswith("."):
            cur.append(l.strip())
    flush()
    common = collections.Counter()
    common_by_op = collections.defaultdict(collections.Counter)
    other = collections.Counter()
    side = []
    listing = []
    for name, lines in blocks:
        sp = side_path(lines)
        if sp:
            side.append((name, sp, len([l for l in lines if l.split()[0].startswith("v_")])))
            continue
        for l in lines:
            l = l.split(";")[0].strip()
            if not l:
                continue
            op, _, operands = l.partition(" ")
            if op.startswith("v_"):
                c = issue_class(op, operands)
                common[c] += 1
                common_by_op[c][re.sub(r"_e(32|64)$", "", op)] += 1
                listing.append((name, c, l))
            else:
                kind = "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "flat_", "buffer_")) else \
                       "wait" if op.startswith("s_waitcnt") or op == "s_nop" else "branch" if op.startswith(("s_cbranch", "s_branch")) else "SALU"
                other[kind] += 1
                listing.append((name, kind, l))
    total = sum(common.values())
    cyc = sum(CYCLES[c] * n for c, n in common.items())
    print("# Point loop of `%s`: per-opcode census\n" % kernel)
    print("Innermost loop header `%s` (depth %d). Common path = every basic block of the loop except the side paths listed below." % (header, top))
    print("(Escape words: each of the three per-symbol escape blocks -- `ds_read_b32` + `v_add_u32` under `s_and_saveexec` -- is counted; a wave")
    print("of the benchmark stream executes them with at least one lane active in 81 % of its symbol steps.)\n")
    print("| issue class | VALU instructions per point | cycles each | cycles per point and wave |")
    print("|---|---|---|---|")
    for c in ("fast", "slow", "rcp"):
        print("| %s | %d | %.1f | %.1f |" % (c, common[c], CYCLES[c], CYCLES[c] * common[c]))
    print("| **all VALU** | **%d** | %.2f on average | **%.1f** |\n" % (total, cyc / max(1, total), cyc))
    print("Other instructions of the common path: " + ", ".join("%d %s" % (n, k) for k, n in sorted(other.items())) + ".\n")
    for c in ("slow", "fast", "rcp"):
        print("**%s**: " % c + ", ".join("`%s` x %d" % (op, n) for op, n in common_by_op[c].most_common()) + "\n")
    print("Side paths inside the loop (not executed by a wave of the benchmark frame, or skipped with an empty exec mask):\n")
    for name, sp, n in side:
        print("* `%s`: %s -- %d VALU" % (name, sp, n))
    print("\n## Listing of the common path (block, class, instruction)\n\n```")
    for name, c, l in listing:
        print("%-10s %-6s %s" % (name, c, l))
    print("```")


if __name__ == "__main__":
    main()
