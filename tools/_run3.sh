timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_a
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_a/**/*kernel_stats.csv', recursive=True)[0]
print({r['Name'].split('(')[0].replace('void fdh::',''): round(float(r['AverageNs'])/1000,1) for r in csv.DictReader(open(f)) if 'blur' in r['Name']})
PY
cd $GRAFT_REPO_ROOT; python bench.py --steps 100 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['one_frame_at_a_time']['value'], d['cpu_baseline']['parity_max_lsb'], d['cpu_baseline']['parity_pixels_differing'])"
