#!/bin/bash
# GPU-box soak of the cross-context check (tools/race_probe.py): several contexts replay different frames with a full-frame
# matrix-pipe blur on their own streams; every context must end with the frame it renders alone.  Sizes / context counts /
# iteration counts from the arguments: race_soak.sh <out.txt> <library or ""> "<WxH:NC:iters:MODE> ..."
out=$1; lib=$2; shift 2
export PYTHONUNBUFFERED=1
[ -n "$lib" ] && export FIGDRAW_HIP_LIB=$lib
lint=$(python3 tools/lint_isa.py ${lib:-figdraw_amd/libfigdraw_hip.so} --allow-packed); lint_rc=$?
echo "library: ${lib:-figdraw_amd/libfigdraw_hip.so}  ($(echo "$lint" | tail -1))  lint_isa exit status: $lint_rc" >> $out
[ $lint_rc -ne 0 ] && echo "WARNING: lint_isa failed for this library (exec-mask write in a uniform build, or a kernel missing): results below are for a build the product Makefile would reject" | tee -a $out
for spec in $*; do
  IFS=: read size nc iters mode <<< "$spec"
  t0=$(date +%s)
  r=$(SIZE=$size NC=$nc MODE=$mode VERBOSE=0 timeout 3000 python3 tools/race_probe.py $iters | tail -1)
  echo "$r  ($(( $(date +%s) - t0 )) s)" | tee -a $out
done
