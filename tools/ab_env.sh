#!/bin/bash
# GPU-box helper: bench.py with an environment switch on and off, same box, interleaved.  usage: bash tools/ab_env.sh FDH_SOMETHING [on off]
var=$1; on=${2:-1}; off=${3:-0}
for rep in 1 2; do for v in $on $off; do
  printf "%s=%s  " $var $v
  env $var=$v timeout 300 python3 bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'replay', d['replay_resident_records']['value'], 'one at a time', d['one_frame_at_a_time']['ms_per_step'], d['one_frame_at_a_time']['batches_ms'][:3], 'diff', d['frames_in_flight_check']['pixels_differing'])"
done; done
