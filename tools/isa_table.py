#!/usr/bin/env python3
"""Print an isa_budget.py JSON as a table."""
import json, sys
d = json.load(open(sys.argv[1]))
print({k: v for k, v in d.items() if k not in ('regions', 'whole kernel (static)')})
keys = ['fp64_fma', 'fp64_mul', 'fp64_add', 'fp64_rcp_rsq_sqrt', 'fp64_div_helpers', 'fp64_other', 'fp32', 'compare', 'select', 'int_mul',
        'int_other', 'mov', 'VALU_total', 'salu', 'branch', 'smem', 'lds', 'vmem', 'wait_nop', 'all']
print('%-30s' % 'region', ' '.join('%5s' % k.replace('fp64_', 'd_')[:5] for k in keys))
for r, c in list(d['regions'].items()) + [('whole', d['whole kernel (static)'])]:
    print('%-30s' % r[:30], ' '.join('%5d' % c.get(k, 0) for k in keys), {k: v for k, v in c.items() if k.startswith('other')})
