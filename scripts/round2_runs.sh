#!/bin/bash
# Round-2 measurement runs (GPU box): lossless workload at two set sizes, table-size sensitivity of the VarDCT bench.
# usage: bash scripts/round2_runs.sh   (writes gpurun_out/lossless_*.json, gpurun_out/sens_*.json and prints one line each)
mkdir -p gpurun_out
for b in 192 384; do
  timeout -k 10 500 python3 bench.py --workload lossless --batch $b --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/lossless_$b.json 2> gpurun_out/lossless_$b.err
  tail -1 gpurun_out/lossless_$b.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lossless', $b, d['value'], d['ms_per_step'], d['roofline']['launch_ms'])" || tail -3 gpurun_out/lossless_$b.err
done
i=0
for spec in "--max-clusters 128" "--max-clusters 128 --distance 0.5" "--distance 0.5"; do
  i=$((i+1))
  timeout -k 10 500 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --e2e-frames 0 $spec > gpurun_out/sens_$i.json 2> gpurun_out/sens_$i.err
  tail -1 gpurun_out/sens_$i.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['config']['entropy_tables']
print('sens', '$spec', d['value'], d['ms_per_step'], d['config']['bpp'], [(x['clusters'], x['log_alpha'], x['lds_bytes'], x['workgroups_per_cu']) for x in t], d['stage_ms_per_frame'])" || tail -3 gpurun_out/sens_$i.err
done
