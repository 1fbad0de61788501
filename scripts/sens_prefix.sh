#!/bin/bash
# Measurement aid (GPU box): prefix-coded AC streams on the lane kernel's prefix form and on the wave-per-section kernel.
python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "prefix or reference_vardct_alpha or outside" > gpurun_out/t6.txt 2>&1; tail -3 gpurun_out/t6.txt
i=0
for spec in "--ac-code-mode 1" "JXLHIP_NO_LANE_PREFIX=1 --ac-code-mode 1 --batch 64 --steps 2"; do
  i=$((i+1))
  envs=""; args=""
  for w in $spec; do case $w in *=*) envs="$envs $w";; *) args="$args $w";; esac; done
  env $envs timeout -k 10 400 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --e2e-frames 0 $args > gpurun_out/sens3_$i.json 2> gpurun_out/sens3_$i.err
  tail -1 gpurun_out/sens3_$i.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('prefix', '$spec', d['value'], d['ms_per_step'], d['config']['bpp'], d['stage_ms_per_frame'])" || tail -3 gpurun_out/sens3_$i.err
done
