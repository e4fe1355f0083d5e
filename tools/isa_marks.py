#!/usr/bin/env python3
"""Dev tool: static instruction counts between the ; ORT_STAGE_* / ORT_FN_* comment markers of one kernel in the
listing of `make -C opticalraytrace_amd/csrc isa`.   usage: python tools/isa_marks.py <kernel name substring>"""
import re, collections, sys
lines = open('build/isa/ort_hip_mark.s').read().splitlines()
pat = sys.argv[1]
s = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l)][0]
e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
CLS = r"(v_cndmask|v_cmp|v_mov|v_readlane|v_writelane|v_fma|v_mul_f64|v_add_f64|v_div|v_rcp|v_rsq|v_mad_u64|v_mul_lo|v_mul_hi|v_lshl|v_lshr|v_and|v_xor|v_or|v_add_co|v_addc|v_add_u|v_sub|v_bfe|v_cvt|scratch|global_load|global_store|ds_|s_load|s_mov|s_cbranch|s_and_saveexec|s_waitcnt|s_nop)"
def count(a, b):
    c = collections.Counter()
    for l in lines[a:b]:
        m = re.match(r"\s+([a-z][a-z_0-9]+)\b", l)
        if not m or l.strip().startswith((".", ";")): continue
        op = m.group(1)
        c['VALU' if op.startswith('v_') else 'SALU' if op.startswith('s_') else 'MEM'] += 1
        mm = re.match(CLS, op)
        c[mm.group(1) if mm else 'other:' + op] += 1
    return c
marks = [(i, l.strip()) for i, l in enumerate(lines[s:e], s) if 'ORT_STAGE_' in l or 'ORT_FN_' in l]
open_ = {}
for i, l in marks:
    kind, name = l.split()[1], l.split()[2]
    if kind.endswith('BEGIN'): open_[name] = i
    elif name in open_:
        c = count(open_[name], i)
        print(f"{name:10s} VALU {c['VALU']:5d} SALU {c['SALU']:5d} MEM {c['MEM']:4d} | " + ", ".join(f"{k} {v}" for k, v in c.most_common() if k not in ('VALU', 'SALU', 'MEM'))[:400])
c = count(s, e)
print(f"{'whole':10s} VALU {c['VALU']:5d} SALU {c['SALU']:5d} MEM {c['MEM']:4d}")
