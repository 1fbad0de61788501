#!/bin/bash
# Round-4 opening measurements at the round-3 HEAD (GPU box): the default bench line, the same with raised wave priority on
# the lane kernel, and the SQ counters of the three hot kernels alone.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04a
mkdir -p $R/$OUT
cd $R
B="python3 bench.py --no-cpu-baseline --e2e-frames 0"
$B > $OUT/bench_default.json 2> $OUT/bench_default.err && tail -c 1500 $OUT/bench_default.json &&
JXLHIP_LANES_PRIO=1 $B > $OUT/bench_prio.json 2> $OUT/bench_prio.err && tail -c 600 $OUT/bench_prio.json &&
$B --no-pipeline > $OUT/bench_nopipe.json 2> $OUT/bench_nopipe.err && tail -c 1500 $OUT/bench_nopipe.json &&
JXLHIP_ENTROPY_GATE=0 timeout -k 10 500 bash scripts/pmc_entropy.sh $OUT/pmc --e2e-frames 0
