#!/usr/bin/env python3
"""Instruction budget of a kernel from the compiler's listing.

    make -C opticalraytrace_amd/csrc isa            # build/isa/ort_hip_mark.s (-DORT_ISA_MARKERS)
    python tools/isa_budget.py build/isa/ort_hip_mark.s 'trace_queue_kernel<0, true, false, double, 1, false>' [OUT.json]

Static counts per region of the listing.  The marker build emits `; ORT_STEP_END k` comment lines
behind every surface step of a program kernel; pure arithmetic is free to move across such a
line, and in practice a step's instructions end up in front of its markers, so region "step k" =
the lines between the END markers of steps k-1 and k (approximate at the edges: emission and step
0 share the first region, queue traffic and loop control sit in the regions around the queue
point).  A step is executed by every ray that reaches it, so static count x rays entering the step
(SURVEY §6 stage-survival table, or the kernel's own counters) is the dynamic budget; the whole-
kernel totals are exact.  Classes follow what costs differently on
gfx950 (profiles/r01/ubench*.log): fp64 fma/mul/add ~1 issue slot, v_rcp/v_rsq_f64 ~3.5 slots,
compares, selects, 32/64-bit integer (the RNG), everything scalar."""
import collections
import json
import re
import subprocess
import sys

CLASSES = [
    ("fp64_fma", r"v_(fma|fmac)_f64"),
    ("fp64_mul", r"v_mul_f64"),
    ("fp64_add", r"v_add_f64"),
    ("fp64_rcp_rsq_sqrt", r"v_(rcp|rsq|sqrt)_f64"),
    ("fp64_div_helpers", r"v_div_(scale|fmas|fixup)_f64"),
    ("fp64_other", r"v_(max|min|floor|rndne|trunc|ldexp|frexp\w*|cvt\w*)_f64|v_cvt_\w+_f64|v_cvt_f64_\w+"),
    ("fp32", r"v_\w+_f32|v_cvt_f32_\w+|v_cvt_\w+_f32"),
    ("compare", r"v_cmp\w*|v_cmpx\w*"),
    ("select", r"v_cndmask_b32"),
    ("int_mul", r"v_mul_(lo|hi)_u32|v_mad_u64_u32|v_mul_u32_u24|v_mad_u32_u24"),
    ("int_other", r"v_(xor|and|or|not|lshl|lshr|ashr|add|sub|alignbit|bfe|bfi|perm|xad|mbcnt|add3|lshl_add|lshl_or|and_or|or3|xor3)\w*"),
    ("mov", r"v_mov_b(32|64)\w*|v_accvgpr\w*|v_readfirstlane\w*|v_readlane\w*|v_writelane\w*"),
    ("lds", r"ds_\w+"),
    ("vmem", r"(global|flat|buffer|scratch)_\w+"),
    ("salu", r"s_(?!waitcnt|nop|endpgm|barrier|cbranch|branch|load|setpc|sleep|buffer)\w+"),
    ("smem", r"s_(load|buffer_load)\w+"),
    ("branch", r"s_cbranch\w+|s_branch|s_setpc\w+"),
    ("wait_nop", r"s_waitcnt\w*|s_nop|s_barrier|s_sleep"),
]
VALU = {"fp32", "fp64_fma", "fp64_mul", "fp64_add", "fp64_rcp_rsq_sqrt", "fp64_div_helpers", "fp64_other", "compare",
        "select", "int_mul", "int_other", "mov"}


def classify(op):
    op = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    for name, pat in CLASSES:
        if re.fullmatch(pat, op):
            return name
    return "other:" + op


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    path, want = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    starts = [(i, m.group(1)) for i, ln in enumerate(lines) if (m := re.match(r"^(_Z\w+):", ln))]
    names = demangle([n for _, n in starts])
    pick = [(i, n) for i, n in starts if want in names[n]]
    if len(pick) != 1:
        sys.exit(f"{len(pick)} kernels match {want!r}: " + "; ".join(names[n][:100] for _, n in pick))
    i0 = pick[0][0]
    i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    regions = collections.OrderedDict()
    cur = "prologue + emission + step 0"
    for ln in lines[i0:i1 + 1]:
        t = ln.strip()
        if re.match(r"; ORT_STEP_BEGIN (\d+)", t):
            continue
        if m := re.match(r"; ORT_STEP_END (\d+)", t):
            cur = f"step {int(m.group(1)) + 1} (+ what follows step {m.group(1)})"
            continue
        if not t or t.startswith((";", ".", "//")) or t.split(";")[0].strip().endswith(":"):
            continue
        op = t.split()[0]
        regions.setdefault(cur, collections.Counter())[classify(op)] += 1
    res = {"kernel": names[pick[0][1]].split("(")[0], "listing": path, "regions": {}}
    meta = "\n".join(lines[i1:i1 + 80])
    for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
        if m := re.search(rf"; {key}: (\d+)", meta):
            res[key] = int(m.group(1))
    total = collections.Counter()
    for r, c in regions.items():
        d = dict(sorted(c.items()))
        d["VALU_total"] = sum(v for k, v in c.items() if k in VALU)
        d["all"] = sum(c.values())
        res["regions"][r] = d
        total.update(c)
    t = dict(sorted(total.items()))
    t["VALU_total"] = sum(v for k, v in total.items() if k in VALU)
    t["all"] = sum(total.values())
    res["whole kernel (static)"] = t
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
