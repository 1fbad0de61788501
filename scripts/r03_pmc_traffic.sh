#!/bin/bash
# HBM traffic per stage kernel (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes of the default bench command, two
# timed steps). The entropy gate is switched off for these passes: counter collection runs the kernels one at a time, and a
# launch that waits for another one's workgroups to be resident has nothing to wait for then. usage (GPU box): this script
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03/profiles
mkdir -p $OUT
( while true; do date >> $OUT/heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB" EXIT
RAW=/tmp/pmc_traffic_$$
mkdir -p $RAW
export TMPDIR=/tmp JXLHIP_ENTROPY_GATE=0
cd /tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-frames 0"
RE="k_entropy|k_idct|k_filter|k_dct|k_special|k_color"
timeout -k 10 ${PMC_TIMEOUT:-260} rocprofv3 --output-format csv --kernel-include-regex "$RE" --pmc FETCH_SIZE -d $RAW/f -o f -- $BENCH > $OUT/pmc_bench.json 2> $RAW/f.log; echo "fetch pass rc $?"
timeout -k 10 ${PMC_TIMEOUT:-260} rocprofv3 --output-format csv --kernel-include-regex "$RE" --pmc WRITE_SIZE -d $RAW/w -o w -- $BENCH > /dev/null 2> $RAW/w.log; echo "write pass rc $?"
FRAMES=$(python3 -c "import json,sys; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['config']['frames_per_step_per_gpu'])" $OUT/pmc_bench.json)  # (what the bench line of this very pass says: not a literal)
python3 $R/scripts/pmc_summary.py $(find $RAW/f -name "*counter_collection.csv" | head -1) $(find $RAW/w -name "*counter_collection.csv" | head -1) $OUT/r03_pmc_traffic.json $FRAMES && head -c 600 $OUT/r03_pmc_traffic.json
tail -2 $RAW/f.log | cut -c1-300
rm -rf $RAW
