#!/bin/bash
# GPU-box helper: per-kernel times (rocprofv3 --stats, one frame at a time) of several builds on the SAME box.
# usage: bash tools/ab_prof.sh name...   (name = tree | build/libfigdraw_hip_<name>.so)
root=$(pwd); cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  lib=$root/build/libfigdraw_hip_$v.so; [ $v = tree ] && lib=$root/figdraw_amd/libfigdraw_hip.so
  rm -rf /tmp/prof_$v
  FIGDRAW_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o r -- python3 $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/prof_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True)[0]
print(sys.argv[1], {r['Name'].split('(')[0].replace('void fdh::',''): round(float(r['AverageNs'])/1000,1) for r in csv.DictReader(open(f)) if 'blur' in r['Name'] or 'composite' in r['Name']})
PY
done; done
