#!/bin/bash
# GPU-box helper: VALU / SALU wave-instruction counts per launch of the bench frame's kernels for one or more builds.
# usage: bash tools/pmc_quick.sh [lib.so ...]   (default: the working tree's library)
root=$(pwd); export TMPDIR=/tmp
libs=${@:-$root/figdraw_amd/libfigdraw_hip.so}
for lib in $libs; do
  out=$root/gpurun_out/pmcq_$(basename $lib .so); rm -rf $out; mkdir -p $out
  (cd /tmp && FIGDRAW_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out -o run -- python3 $root/tools/one_frame.py > $out.log 2>&1)
  f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
  echo "== $(basename $lib)"; [ -n "$f" ] && python3 $root/tools/pmc_summary.py $f | grep -A3 "k_composite_tiles<4, true>\|k_bin_draws"
done
