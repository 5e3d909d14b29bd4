#!/bin/bash
# GPU-box helper: per-dispatch trace of a short one-frame-at-a-time run, reduced to the gap in front of every kernel of a frame (median over frames)
root=$(pwd); export TMPDIR=/tmp; out=$root/gpurun_out/trace_gaps; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o r -- python3 $root/bench.py --steps 40 --warmup 5 --repeats 1 --frames-in-flight 1 --no-cpu-baseline > /dev/null 2>&1)
python3 - $(ls $out/*kernel_trace.csv $out/*/*kernel_trace.csv 2>/dev/null | head -1) <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    k = b["Kernel_Name"].split("(")[0][-40:]
    gaps[k].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1000); durs[k].append((int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1000)
for k in gaps:
    g = sorted(gaps[k]); d = sorted(durs[k])
    print(f"{k:42s} n {len(g):4d}  gap in front: median {g[len(g)//2]:6.2f} us   duration median {d[len(d)//2]:6.2f} us")
PY
