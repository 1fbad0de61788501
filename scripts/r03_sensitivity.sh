#!/bin/bash
# Stream sensitivity of the headline at HEAD: the default bench command on other
# kinds of stream. One line per run in gpurun_out/r03/r03_sensitivity.txt. usage (GPU box): bash scripts/r03_sensitivity.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
: > $OUT/r03_sensitivity.txt
for spec in "--distance 0.5" "--max-clusters 128" "--max-clusters 128 --distance 0.5" "--ac-code-mode 1" "--distance 2.0"; do
  timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --e2e-frames 0 $spec > $OUT/sens_last.json 2> $OUT/sens_last.err
  tail -1 $OUT/sens_last.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
t=d['config'].get('entropy_tables') or []
print(json.dumps({'args': '$spec', 'MP/s': d['value'], 'ms_per_step': d['ms_per_step'], 'bpp': d['config'].get('bpp'), 'stage_ms_per_frame': d['stage_ms_per_frame'],
                  'entropy_launch_ms_alone': d['roofline'].get('launch_ms_alone'), 'tables': t[:2]}))" >> $OUT/r03_sensitivity.txt || tail -3 $OUT/sens_last.err >> $OUT/r03_sensitivity.txt
  tail -1 $OUT/r03_sensitivity.txt | cut -c1-300
done
timeout -k 10 300 python3 bench.py --workload lossless --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r03_lossless_bench.json 2> $OUT/sens_last.err
tail -1 $OUT/r03_lossless_bench.json | cut -c1-400
