#!/bin/bash
# End-to-end pipeline variants (GPU box): chunk size and mover threads of bench.py's e2e block. usage: r04_e2e_variants.sh OUTDIR
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/e2e}
mkdir -p $R/$OUT
cd $R
for v in "256 4" "128 4" "192 4" "256 6" "128 6"; do
  set -- $v
  JXLAMD_E2E_CHUNK=$1 JXLAMD_E2E_MOVERS=$2 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-libjxl-tables 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read())['e2e']; print('chunk $1 movers $2:', d['value'], 'MP/s', d['ms_per_frame'], d['stage_busy_ms_per_frame'])" | tee -a $OUT/e2e_variants.txt
done
