#!/bin/bash
# GPU-box helper: blocks per wave of the fused blur kernel (FDH_FX_T; default: the smallest T with every wave resident, 5 at 4K), four contexts
# in flight and one frame at a time
for rep in 1 2; do for t in 0 4 6 7 9 12; do
  printf "FDH_FX_T=%-3s " $t
  if [ $t = 0 ]; then unset FDH_FX_T; else export FDH_FX_T=$t; fi
  timeout 300 python3 bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'replay', d['replay_resident_records']['value'], 'one at a time', d['one_frame_at_a_time']['ms_per_step'], 'fused ms', d['roofline_blur']['fused_route']['ms'], 'diff', d['frames_in_flight_check']['pixels_differing'])"
done; done
