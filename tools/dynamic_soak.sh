#!/bin/bash
# usage (GPU box): bash tools/dynamic_soak.sh out.txt [bursts]
out=$1; n=${2:-150}
echo "library: figdraw_amd/libfigdraw_hip.so  ($(python3 tools/lint_isa.py figdraw_amd/libfigdraw_hip.so | grep -o 'packed-FP32 instructions: [0-9]*' | head -1))" >> $out
for size in 1280x720 1920x1080 3840x2160; do
  for spec in "4 1 -1" "4 4 -1" "4 1 1" "4 4 1" "3 3 0" "6 2 -1"; do
    set -- $spec
    t0=$(date +%s)
    r=$(SIZE=$size NC=$1 THREADS=$2 ROUTE=$3 timeout 1200 python3 tools/dynamic_soak.py $n 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3 | tr '\n' ' ')
    echo "$r ($(( $(date +%s) - t0 )) s)" | tee -a $out
  done
done
