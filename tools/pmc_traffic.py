#!/usr/bin/env python3
"""Assemble profiles/<tag>_pmc_traffic.json from rocprofv3 counter_collection CSVs (see tools/pmc_traffic.sh)."""
import collections, csv, json, sys

tag, files = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("fdh::"):
            continue
        agg[(k[5:], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
# name the launches of a frame: the largest grid of each kernel is the full-frame / phase-0 launch
by_kernel = collections.defaultdict(list)
for (k, g) in agg:
    by_kernel[k.split("<")[0]].append((g, k))
kernels = {}
for base, lst in by_kernel.items():
    lst.sort(reverse=True)
    for i, (g, k) in enumerate(lst):
        c = {n: sum(v) / len(v) for n, v in agg[(k, g)].items()}
        if base == "k_composite_tiles":
            label = "k_composite_tiles.phase0" if i == 0 else f"k_composite_tiles.later{i}"
        elif base == "k_blur_mx":  # k_blur_mx<NK, false> = horizontal pass, <NK, true> = vertical pass
            label = ("k_blur_mx.v" if "true" in k else "k_blur_mx.h") + ("" if sum(1 for g2, k2 in lst[:i] if ("true" in k2) == ("true" in k)) == 0 else f".{i}")
        elif base == "k_blur_fx":
            label = "k_blur_fx" if i == 0 else f"k_blur_fx.{i}"
        elif base in ("k_blur_h", "k_blur_v"):
            label = f"{base}.largest" if i == 0 else f"{base}.{i}"
        else:
            label = base if i == 0 else f"{base}.{i}"
        e = {"kernel": k, "grid": g}
        e.update({n: round(v, 1) for n, v in c.items()})
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["hbm_bytes"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
        kernels[label] = e
import hashlib, os
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "figdraw_amd", "libfigdraw_hip.so")
lib_sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None  # which build was profiled
print(json.dumps({"_about": "rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum / SQ_INSTS_VALU SQ_INSTS_SALU ... in separate runs, each with "
                  "--kernel-trace only) over `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline`, MI355X, S300@4K frame. Means per "
                  "dispatch in the counters' native unit (KB for *_SIZE). hbm_bytes applies MI355X_MICROARCH.md's gfx950 correction: "
                  "FETCH_SIZE counts 128-B requests as 64 B, so reads are doubled: (2*FETCH_SIZE + WRITE_SIZE) * 1024.",
                  "tag": tag, "library_sha16": lib_sha, "kernels": kernels}, indent=1))
