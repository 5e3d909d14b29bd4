#!/bin/bash
# GPU-box helper: VALU wave-instructions of the bench frame's kernels BY CLASS (the SQ's own per-type counters), two passes.
#   FMA_F32 / MUL_F32 / ADD_F32: the 2.4-cycle class (when no operand is an SGPR); TRANS_F32: v_exp / v_rcp / v_sqrt (8 cycles);
#   CVT: byte <-> float conversions; INT32: integer adds / shifts / logic;  the REST of SQ_INSTS_VALU is select / min / max / compare /
#   round / move: the 4-cycle class ("What the VALU gives").
# usage: bash tools/pmc_classes.sh [lib.so ...]    (default: the working tree's library)
root=$(pwd); export TMPDIR=/tmp
libs=${@:-$root/figdraw_amd/libfigdraw_hip.so}
for lib in $libs; do
  echo "== $(basename $lib)"
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU" \
             "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
    i=$((i+1))
    out=$root/gpurun_out/pmcc_$(basename $lib .so)_$i; rm -rf $out; mkdir -p $out
    (cd /tmp && FIGDRAW_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o run -- python3 $root/tools/one_frame.py > $out.log 2>&1)
    f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 $root/tools/pmc_summary.py $f | grep -A9 "k_composite_tiles<4, true>\|k_blur_fx"
  done
done
