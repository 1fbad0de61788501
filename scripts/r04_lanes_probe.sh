#!/bin/bash
# Lane-kernel probes of round 4 (GPU box): the isolated trip loop, and the 640-frame launch alone under several settings
# (refill cadence, pass threshold), each also with the kernel's own cycle split. usage: bash scripts/r04_lanes_probe.sh OUTDIR [CFG ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/r04h}
shift
mkdir -p $R/$OUT
cd $R
bash scripts/ubench_trip3.sh 2000 > $OUT/trip3_variants.txt 2>&1
head -8 $OUT/trip3_variants.txt
CFGS=("$@")
if [ ${#CFGS[@]} -eq 0 ]; then CFGS=("JXLHIP_REFILL_EVERY=1" "JXLHIP_REFILL_EVERY=2" "JXLHIP_REFILL_EVERY=4" "JXLHIP_LANES_CPP=1"); fi
python3 scripts/r03_entropy_probe.py 640 "${CFGS[@]}" 2>&1 | grep "entropy" > $OUT/alone.txt
cat $OUT/alone.txt
PCFGS=()
for c in "${CFGS[@]}"; do PCFGS+=("JXLHIP_LANES_PROF=1,$c"); done
python3 scripts/r03_entropy_probe.py 640 "${PCFGS[@]}" 2>&1 | grep "lanes prof\|entropy" | awk '/lanes prof/{last=$0} /entropy/{print last; print}' > $OUT/split.txt
cat $OUT/split.txt | cut -c1-700
