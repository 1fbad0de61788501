#!/bin/bash
# The counter / probe part of the round-3 profile set on its own (run on the GPU box after scripts/profile_round.sh's kernel
# statistics are in): HBM traffic per kernel, the lane kernel's cycle split and SQ counters, isolated stage launches.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03/profiles
mkdir -p $R/$OUT
cd $R
( while true; do date >> $R/$OUT/heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB" EXIT
RAW=/tmp/prof_raw_$$
mkdir -p $RAW
export TMPDIR=/tmp
export JXLHIP_ENTROPY_GATE=0  # (counter collection serialises the kernels: a gated launch would wait for one that cannot start)
PMCBENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-frames 0"
( cd /tmp && timeout -k 10 300 rocprofv3 --output-format csv --kernel-include-regex "k_entropy|k_idct|k_filter|k_dct|k_special|k_color" --pmc FETCH_SIZE -d $RAW/f -o f -- $PMCBENCH > $R/$OUT/pmc_bench.json 2> $RAW/f.log )
( cd /tmp && timeout -k 10 300 rocprofv3 --output-format csv --kernel-include-regex "k_entropy|k_idct|k_filter|k_dct|k_special|k_color" --pmc WRITE_SIZE -d $RAW/w -o w -- $PMCBENCH > /dev/null 2> $RAW/w.log )
FRAMES=$(python3 -c "import json,sys; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['config']['frames_per_step_per_gpu'])" $R/$OUT/pmc_bench.json)  # (what the bench line of this very pass says: not a literal)
python3 $R/scripts/pmc_summary.py $(find $RAW/f -name "*counter_collection.csv" | head -1) $(find $RAW/w -name "*counter_collection.csv" | head -1) $R/$OUT/r03_pmc_traffic.json $FRAMES
tail -3 $RAW/f.log
JXLHIP_LANES_PROF=1 python3 scripts/r03_entropy_probe.py 640 base 2>&1 | grep "lanes prof" | tail -1 | sed 's/^\[lanes prof\] //' > $R/$OUT/r03_entropy_split.json
python3 scripts/r03_stage_times.py 256 base 2>&1 | tail -4 > $R/$OUT/r03_stage_times.txt
timeout -k 10 420 bash scripts/r03_pmc_probe.sh $OUT/pmc 640 > /dev/null 2>&1
cp $R/$OUT/pmc/summary.txt $R/$OUT/r03_sq_counters_entropy.txt
rm -rf $RAW
ls -la $R/$OUT
