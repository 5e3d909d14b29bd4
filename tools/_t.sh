root=$(pwd); cd /tmp; export TMPDIR=/tmp
for t in 1 2 3 4 5 6 8 12; do
  rm -rf /tmp/prof_t
  FDH_MX_T=$t rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o r -- python3 $root/bench.py --steps 40 --warmup 5 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2>&1
  python3 - $t <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/prof_t/**/*kernel_stats.csv', recursive=True)[0]
print("T =", sys.argv[1], {r['Name'].split('(')[0].replace('void fdh::',''): round(float(r['AverageNs'])/1000,1) for r in csv.DictReader(open(f)) if 'blur_mx' in r['Name']})
PY
done
