#!/bin/bash
# GPU-box helper: tools/thread_churn.py N times, printing the runs that are not clean.  usage: [ENV=..] bash tools/churn_loop.sh N [threads contexts]
n=0; N=${1:-50}
for i in $(seq 1 $N); do
  out=$(timeout 200 python3 tools/thread_churn.py ${2:-4} ${3:-8} 2>&1)
  if ! echo "$out" | grep -q ": 0 wrong"; then n=$((n+1)); echo "RUN $i:"; echo "$out" | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|coredump\|core dump\|segment data" | tail -4; fi
done
echo "not clean: $n of $N"
