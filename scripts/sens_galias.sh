python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "alias_tables_in_global or batched_entropy or fallback or lane_packing" > gpurun_out/t5.txt 2>&1; tail -3 gpurun_out/t5.txt
i=0
for spec in "--max-clusters 128" "JXLHIP_GALIAS=0 --max-clusters 128" "--max-clusters 128 --distance 0.5" "JXLHIP_GALIAS=1"; do
  i=$((i+1))
  envs=""; args=""
  for w in $spec; do case $w in *=*) envs="$envs $w";; *) args="$args $w";; esac; done
  env $envs timeout -k 10 400 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --e2e-frames 0 $args > gpurun_out/sens2_$i.json 2> gpurun_out/sens2_$i.err
  tail -1 gpurun_out/sens2_$i.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('sens', '$spec', d['value'], d['ms_per_step'], d['stage_ms_per_frame'])" || tail -3 gpurun_out/sens2_$i.err
done
