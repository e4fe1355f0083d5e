#!/usr/bin/env python3
"""Dev tool: instruction-class counts of one kernel in the compiler listing (make -C opticalraytrace_amd/csrc isa).
usage: python tools/isa_func.py <substring of the mangled name> [listing]"""
import collections, re, sys
pat = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "build/isa/ort_hip_mark.s"
lines = open(path).read().splitlines()
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l)]
for s in start:
    e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cnt = collections.Counter()
    for l in lines[s:e]:
        m = re.match(r"\s+([a-z][a-z_0-9]+)\b", l)
        if not m or l.strip().startswith((".", ";")): continue
        op = m.group(1)
        for key in ("scratch_load", "scratch_store", "v_writelane", "v_readlane", "v_mov", "v_cndmask", "s_load", "global_load", "global_store", "global_atomic",
                    "ds_", "v_fma", "v_mul_f64", "v_add_f64", "v_cmp", "v_div", "v_rcp", "v_rsq", "v_sqrt", "s_cbranch", "s_and_saveexec", "s_mov", "v_mad_u64", "v_lshl", "v_and", "v_xor", "v_or"):
            if op.startswith(key): cnt[key] += 1; break
        else:
            cnt["other_" + op.split("_")[0]] += 1
    total = sum(cnt.values())
    print(lines[s][:90], "static instructions", total)
    print("  " + ", ".join(f"{k} {v}" for k, v in cnt.most_common()))
