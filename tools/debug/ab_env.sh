#!/bin/bash
# GPU box: A/B of an environment switch on bench.py, alternating runs.  usage: bash tools/debug/ab_env.sh VAR A B [rounds]
var=$1; a=$2; b=$3; n=${4:-3}
for i in $(seq 1 $n); do
  for v in $a $b; do
    env $var=$v python bench.py --no-cpu-baseline --repeats 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', 'value', d['value'], 'replay', d['replay_resident_records']['value'], 'one-at-a-time', d['one_frame_at_a_time']['ms_per_step'], 'one replay', d['one_frame_at_a_time']['replay_resident_records']['ms_per_step'], 'comp0', d['roofline']['ms_per_launch'])"
  done
done
