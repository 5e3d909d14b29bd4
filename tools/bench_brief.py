#!/usr/bin/env python3
"""one bench.py JSON line -> the few numbers a round's log quotes.  usage: tools/bench_brief.py gpurun_out/x_bench.json [cfg6.json ...]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "threads", d["config"].get("host_threads_per_gpu"), d.get("host_threads_calibration"))
print("  per_call", d["per_call_path"]["value"], "| replay", d["replay_resident_records"]["value"], "| one-at-a-time", d["one_frame_at_a_time"]["value"],
      d["one_frame_at_a_time"]["ms_per_step"], "replay", d["one_frame_at_a_time"]["replay_resident_records"]["ms_per_step"])
dp = d.get("dynamic_path", {})
print("  host us: record", dp.get("host_record_us"), "prepare", dp.get("host_prepare_us"), "issue", dp.get("host_issue_us"), "| retained", {k: v for k, v in dp.get("retained", {}).items() if k != "note"})
print("  roofline", d["roofline"]["ms_per_launch"], d["roofline"]["frac"], "| compositor", (d.get("roofline_compositor") or {}).get("ms_per_launch"), (d.get("roofline_compositor") or {}).get("frac"), "| two-pass blur", d["roofline_blur"]["frac"], d["roofline_blur"]["passes"]["horizontal"]["ms"], d["roofline_blur"]["passes"]["vertical"]["ms"])
for f in sys.argv[2:]:
    for k, e in json.load(open(f)).items():
        print(" ", k, "draws", e["draws"], "frame_us", e["frame_us"], "dynamic_us", e.get("dynamic_us_per_frame"), e["kernel_us"])
