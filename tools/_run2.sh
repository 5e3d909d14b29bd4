cd /tmp && export TMPDIR=/tmp
for v in 1 2 4 8; do
  rm -rf /tmp/prof_$v
  FDH_MX_T=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/prof_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True)[0]
print(sys.argv[1], {r['Name'].split('(')[0].replace('void fdh::',''): round(float(r['AverageNs'])/1000,1) for r in csv.DictReader(open(f)) if '_mx' in r['Name']})
PY
done
