#!/bin/bash
# GPU-box helper: bench every build/libfigdraw_hip_<name>.so given on the command line (experiment builds)
for v in "$@"; do
  echo -n "$v: "
  FIGDRAW_HIP_LIB=build/libfigdraw_hip_$v.so python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['frame']['kernel_ms'])"
done
