#!/bin/bash
# What each path of the compositor's draw loop costs on the bench frame: one variant library per path LEFT OUT (-DFDH_ABLATE_PATHS=bit),
# VALU / SALU wave-instructions and the launch's time per variant.  (The differences are NOT additive and not per-path costs: leaving a path out also
# changes who leads the shared-distance-field runs and what the compiler does with the rest -- round 5 read ~520 instructions per vertex-colour inner
# shadow off them where the timing build says it costs what a gradient fill's edge strip costs, ~4 300 wave-cycles.  Use them to rank, not to budget.)  Build here (no GPU needed), measure on the GPU box:
#   bash tools/ablate_paths.sh build        -> build/libfigdraw_hip_abl<bit>.so
#   bash tools/ablate_paths.sh              -> counts per variant (tools/pmc_quick.sh)
bits="${BITS:-1 2 4 8 16 32 64 127}"
if [ "$1" = build ]; then
  for b in $bits; do make -C figdraw_amd/csrc variant NAME=abl$b DEFS="-DFDH_ABLATE_PATHS=$b" > build/abl$b.log 2>&1 & done; wait
  ls -la build/libfigdraw_hip_abl*.so; exit 0
fi
names=("" "uniform blend (plain core strips)" "packed edge: fill" "packed edge: drop shadow" "packed edge: inner shadow" "packed edge: AA stroke" "generic path, core strips" "generic path, edge strips")
bash tools/pmc_quick.sh $(pwd)/figdraw_amd/libfigdraw_hip.so $(for b in $bits; do echo $(pwd)/build/libfigdraw_hip_abl$b.so; done) | grep "^==\|<4, true>\|INSTS_VALU\|INSTS_SALU"
