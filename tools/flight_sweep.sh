#!/bin/bash
# value / per-call / replay against the number of contexts in flight (bench.py --frames-in-flight)
for f in "$@"; do
  python bench.py --no-cpu-baseline --frames-in-flight $f --repeats 5 2>/dev/null | tail -1 > /tmp/fs.json
  python - "$f" <<'PY'
import json, sys
j = json.load(open("/tmp/fs.json"))
print(sys.argv[1], j["value"], j["per_call_path"]["value"], j["replay_resident_records"]["value"], flush=True)
PY
done
