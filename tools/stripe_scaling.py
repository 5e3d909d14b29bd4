#!/usr/bin/env python3
"""ONE-GPU stand-in for the 1/2/4/8-GPU row-stripe scaling curve of BASELINE config 5 (no 8-GPU node has been available).

For N = 1, 2, 4, 8 it renders, on this one GPU, each of the N stripes of the 8K frame the way rank r of an N-GPU run would
(fdh_set_stripe(fdh_stripe_rows(H, N, r)): the stripe plus its redundant vertical blur halo; fdh_render_frame per frame,
frames_in_flight contexts) and times it.  The PREDICTED frame period of the N-GPU run is then
    max( slowest stripe's time,  stripe bytes / per-link xGMI bandwidth )
-- compute and the gather overlap (the gather of frame k is queued behind frame k on its context's stream while frames
k + 1 .. run), the 7 links into rank 0 carry the 7 remote stripes concurrently (MI355X_MICROARCH.md: 7 links x ~153 GB/s per
GPU, point to point), rank 0's own stripe needs no transfer.  Prediction, not measurement: labelled as such wherever it is quoted.
usage (GPU box): python3 tools/stripe_scaling.py [W H] > gpurun_out/stripe_scaling.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figdraw_amd import call_stream as CS  # noqa: E402
from figdraw_amd.context import HipContext  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100  # noqa: E402
from figdraw_amd.sharding import stripe_rows  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
LINK_GBS = 153.0
PCIE_GBS = 56.0
F, K = 4, 120
scenes = [make_render_tree_100(w, h, frame=f, full_frame_blur=True) for f in range(8)]
cs = [s.to_c() for s in scenes]
P = CS.Player()
ctxs = [HipContext(device=0) for _ in range(F)]
out = {"_about": __doc__.split("\n\n")[0], "frame": [w, h], "frames_in_flight": F, "link_GBs_assumed": LINK_GBS, "worlds": {}}
for world in (1, 2, 4, 8):
    per_rank = []
    for r in range(world):
        y0, y1 = stripe_rows(h, world, r)
        for c in ctxs:
            c.set_stripe(y0, y1)
        P.play_scenes(ctxs, cs, 24, w, h)
        ts = [P.play_scenes(ctxs, cs, K, w, h) / K * 1e6 for _ in range(3)]
        per_rank.append({"rank": r, "rows": [y0, y1], "us_per_frame": round(float(np.median(ts)), 2)})
    rows_remote = max((b - a for a, b in (stripe_rows(h, world, r) for r in range(1, world))), default=0)
    link_us = rows_remote * w * 4 / (LINK_GBS * 1e9) * 1e6
    slowest = max(p["us_per_frame"] for p in per_rank)
    period = max(slowest, link_us)
    # the other consumer (bench.py --mode stripes --gather host): no gather, every rank reads its own stripe back over its own PCIe
    # link (measured on this pool: 132.7 MB in 2.37 ms = 56 GB/s, profiles/r04_stripes_8k_host.json); the links work in parallel
    rows_max = max(b - a for a, b in (stripe_rows(h, world, r) for r in range(world)))
    pcie_us = rows_max * w * 4 / (PCIE_GBS * 1e9) * 1e6
    host_period = max(slowest, pcie_us)
    out["worlds"][str(world)] = {"stripes": per_rank, "slowest_stripe_us": slowest, "sum_of_stripes_us": round(sum(p["us_per_frame"] for p in per_rank), 2),
                                 "gather_us_per_frame_per_link": round(link_us, 2), "predicted_frame_period_us": round(period, 2),
                                 "predicted_mpixels_per_s": round(w * h / period, 1), "bound": "gather (one xGMI link)" if link_us > slowest else "slowest stripe (compute)",
                                 "host_consumer": {"readback_us_per_stripe": round(pcie_us, 2), "predicted_frame_period_us": round(host_period, 2),
                                                   "predicted_mpixels_per_s": round(w * h / host_period, 1),
                                                   "bound": "each rank's PCIe link" if pcie_us > slowest else "slowest stripe (compute)"}}
base = out["worlds"]["1"]["predicted_frame_period_us"]
for k, v in out["worlds"].items():
    v["predicted_speedup"] = round(base / v["predicted_frame_period_us"], 2)
    v["predicted_efficiency"] = round(base / v["predicted_frame_period_us"] / int(k), 3)
    v["halo_and_launch_overhead"] = round(v["sum_of_stripes_us"] / base, 3)  # total GPU time of all stripes / the unstriped frame
    v["host_consumer"]["predicted_speedup"] = round(out["worlds"]["1"]["host_consumer"]["predicted_frame_period_us"] / v["host_consumer"]["predicted_frame_period_us"], 2)
print(json.dumps(out, indent=1))
