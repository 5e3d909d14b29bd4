#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch, grouped by grid size)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not k.startswith('fdh::'):
        continue
    agg[(k, r.get('Grid_Size'), r.get('VGPR_Count'), r.get('SGPR_Count'))][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, 'dispatches=', len(next(iter(v.values()))))
    for c, x in sorted(v.items()):
        print('   %-28s %14.0f' % (c, sum(x) / len(x)))
