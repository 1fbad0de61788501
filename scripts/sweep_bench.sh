#!/bin/bash
# Measurement aid: bench.py over a few settings, one line each (MP/s, ms/step, per-stage ms per frame).
# usage (on the GPU box): bash scripts/sweep_bench.sh OUT.txt "ENV=.. ARGS.." ...
OUT=$1
shift
: > $OUT
for spec in "$@"; do
  envs=""
  args=""
  for w in $spec; do
    case $w in
      *=*) envs="$envs $w" ;;
      *) args="$args $w" ;;
    esac
  done
  line=$(env $envs timeout -k 10 400 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --e2e-frames 0 $args 2>/dev/null | tail -1)
  python3 - "$spec" "$line" >> $OUT <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print("%-60s %9.0f MP/s %8.2f ms/step  %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k.split()[0]: v for k, v in d["stage_ms_per_frame"].items()}))
except Exception as e:
    print("%-60s FAILED %s" % (sys.argv[1], sys.argv[2][:200]))
PY
done
cat $OUT
