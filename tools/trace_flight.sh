#!/bin/bash
# GPU-box helper: per-dispatch trace of a short run with four contexts in flight (bench.py default), reduced to: how much of the timed
# window has 0 / 1 / 2 / 3+ kernels running, and per kernel the median gap to the previous dispatch OF ITS OWN QUEUE.
root=$(pwd); export TMPDIR=/tmp; out=$root/gpurun_out/trace_flight; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o r -- python3 $root/bench.py --steps 200 --warmup 20 --repeats 1 --host-threads 1 --no-cpu-baseline > /dev/null 2>&1)
python3 - $(ls $out/*kernel_trace.csv $out/*/*kernel_trace.csv 2>/dev/null | head -1) <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ups = [i for i, r in enumerate(rows) if "k_upload_frame" in r["Kernel_Name"]]
N = 160  # the densest stretch of N consecutive uploads that involves four queues: the middle of a timed batch
best = None
for a in range(0, len(ups) - N):
    if len(set(rows[i]["Queue_Id"] for i in ups[a:a + N])) < 4: continue
    dt = int(rows[ups[a + N]]["Start_Timestamp"]) - int(rows[ups[a]]["Start_Timestamp"])
    if best is None or dt < best[0]: best = (dt, ups[a], ups[a + N])
dt, i0, i1 = best
t0, t1 = int(rows[i0]["Start_Timestamp"]), int(rows[i1]["Start_Timestamp"])
print("window %.1f us for %d frames -> %.2f us per frame (under the tracer)" % (dt / 1000, N, dt / 1000 / N))
win = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
ev = []
for r in win:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((min(int(r["End_Timestamp"]), t1), -1))
ev.sort()
cur = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[min(cur, 4)] += t - last; last = t; cur += d
tot = sum(hist.values())
print("kernels running at once: " + "  ".join("%d%s: %.1f %%" % (k, "+" if k == 4 else "", 100 * v / tot) for k, v in sorted(hist.items())))
prev = {}; gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for r in win:
    q = r["Queue_Id"]; k = r["Kernel_Name"].split("(")[0][-38:]
    if q in prev: gaps[k].append((int(r["Start_Timestamp"]) - prev[q]) / 1000)
    prev[q] = int(r["End_Timestamp"]); durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k in durs:
    g = sorted(gaps[k]) or [0]; d = sorted(durs[k])
    print(f"{k:40s} n {len(d):4d}  own-queue gap in front: median {g[len(g)//2]:6.2f} mean {sum(g)/len(g):6.2f} us   duration median {d[len(d)//2]:6.2f} mean {sum(d)/len(d):6.2f} us")
print("sum of the kernels' durations per frame: %.1f us" % (sum(sum(d) for d in durs.values()) / N))
PY
