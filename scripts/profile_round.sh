#!/bin/bash
# Round profiles (run on the GPU box): rocprofv3 kernel statistics of the default bench command and of the same frames
# with the stages run one after the other (--no-pipeline: every kernel alone on the GPU), and FETCH_SIZE / WRITE_SIZE per
# kernel in separate --pmc passes. Raw rocprofv3 output stays in /tmp; only the summaries go to OUTDIR (gpurun_out/...),
# from where they are copied into profiles/.
# usage: bash scripts/profile_round.sh OUTDIR TAG [bench args]
OUT=${1:-gpurun_out/prof}
TAG=${2:-r03}
shift 2 || true
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
RAW=/tmp/prof_raw_$$
mkdir -p $RAW
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --e2e-frames 0 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o s -- $BENCH > $R/$OUT/${TAG}_bench.json 2> $RAW/stats.log
cp $(find $RAW/stats -name "*kernel_stats.csv" | head -1) $R/$OUT/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/iso -o s -- $BENCH --no-pipeline > $R/$OUT/${TAG}_bench_isolated.json 2> $RAW/iso.log
cp $(find $RAW/iso -name "*kernel_stats.csv" | head -1) $R/$OUT/${TAG}_kernel_stats_isolated.csv
# (counter passes: fewer steps, the per-launch counters do not depend on the count; a heartbeat file keeps the run from
# looking hung to the GPU box's silence watchdog while rocprofv3 writes only under /tmp)
( while true; do date >> $R/$OUT/heartbeat.txt; sleep 45; done ) &
HB=$!
export JXLHIP_ENTROPY_GATE=0  # (counter collection serialises the kernels: a gated launch would wait for one that cannot start)
PMCBENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-frames 0 $@"
rocprofv3 --output-format csv --kernel-include-regex "k_entropy|k_idct|k_filter|k_dct|k_special|k_color" --pmc FETCH_SIZE -d $RAW/f -o f -- $PMCBENCH > /dev/null 2> $RAW/f.log
rocprofv3 --output-format csv --kernel-include-regex "k_entropy|k_idct|k_filter|k_dct|k_special|k_color" --pmc WRITE_SIZE -d $RAW/w -o w -- $PMCBENCH > /dev/null 2> $RAW/w.log
kill $HB
# frames per launch: what the bench line of this very run says (not a literal: --batch may be among the arguments)
FRAMES=$(python3 -c "import json,sys; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['config']['frames_per_step_per_gpu'])" $R/$OUT/${TAG}_bench.json)
python3 $R/scripts/pmc_summary.py $(find $RAW/f -name "*counter_collection.csv" | head -1) $(find $RAW/w -name "*counter_collection.csv" | head -1) $R/$OUT/${TAG}_pmc_traffic.json $FRAMES
tail -1 $R/$OUT/${TAG}_bench.json | cut -c1-300
head -8 $R/$OUT/${TAG}_kernel_stats.csv | cut -c1-200
rm -rf $RAW
