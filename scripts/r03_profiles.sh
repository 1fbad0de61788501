#!/bin/bash
# Round-3 profile set (run on the GPU box; summaries land under gpurun_out/r03/profiles, to be copied into profiles/):
#   r03_bench.json / r03_kernel_stats.csv                    rocprofv3 --kernel-trace --stats of the default bench command
#   r03_bench_isolated.json / r03_kernel_stats_isolated.csv  the same frames with the stages one after the other
#   r03_pmc_traffic.json                                      FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes)
#   r03_entropy_split.json                                    the lane kernel's own cycle split (per wave: trips / passes / refills)
#   r03_sq_counters_entropy.txt                               SQ / GRBM counters of the lane kernel alone
#   r03_streams_probe.txt                                     launch time of one 8-frame entropy launch vs HIP streams alive
#   r03_stage_times.txt                                       transform / filter launches alone (every launch listed)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03/profiles
mkdir -p $R/$OUT
cd $R
bash scripts/profile_round.sh $OUT r03 > $R/$OUT/profile_round.log 2>&1
( while true; do date >> $R/$OUT/heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB" EXIT
JXLHIP_LANES_PROF=1 python3 scripts/r03_entropy_probe.py 640 base 2>&1 | grep "lanes prof" | tail -1 | sed 's/^\[lanes prof\] //' > $R/$OUT/r03_entropy_split.json
bash scripts/r03_pmc_probe.sh $OUT/pmc 640 > /dev/null 2>&1
cp $R/$OUT/pmc/summary.txt $R/$OUT/r03_sq_counters_entropy.txt
JXLHIP_LANES=64 JXLHIP_WPG=1 python3 scripts/r03_streams_probe2.py 2>&1 | grep entropy > $R/$OUT/r03_streams_probe.txt
python3 scripts/r03_stage_times.py 256 base 2>&1 | tail -4 > $R/$OUT/r03_stage_times.txt
ls -la $R/$OUT
