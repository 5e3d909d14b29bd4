#!/usr/bin/env python3
"""Per-frame kernel time vs wall time from a rocprofv3 --kernel-trace CSV (start/end timestamps): how much of a frame
is launch gaps.  usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'fdh::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# frames start at k_bin_draws
frames, cur = [], []
for r in rows:
    if 'k_bin_draws' in r['Kernel_Name'] and cur:
        frames.append(cur); cur = []
    cur.append(r)
frames.append(cur)
frames = [f for f in frames if len(f) == len(frames[len(frames) // 2])][2:]
busy = [sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in f) for f in frames]
span = [int(f[-1]['End_Timestamp']) - int(f[0]['Start_Timestamp']) for f in frames]
period = [int(b[0]['Start_Timestamp']) - int(a[0]['Start_Timestamp']) for a, b in zip(frames, frames[1:])]
print(f"frames {len(frames)}  kernels/frame {len(frames[0])}  busy {sum(busy)/len(busy)/1e3:.1f} us  span {sum(span)/len(span)/1e3:.1f} us  "
      f"period {sum(period)/max(len(period),1)/1e3:.1f} us")
per = collections.OrderedDict()
for f in frames:
    for i, r in enumerate(f):
        k = (i, r['Kernel_Name'].split('(')[0].replace('void ', ''))
        per.setdefault(k, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        if i + 1 < len(f):
            per.setdefault((i, '  gap after'), []).append(int(f[i + 1]['Start_Timestamp']) - int(r['End_Timestamp']))
for k, v in per.items():
    print(f"  {k[0]:2d} {k[1]:40s} {sum(v)/len(v)/1e3:8.2f} us")
