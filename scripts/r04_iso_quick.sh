#!/bin/bash
# Quick isolated kernel statistics (run on the GPU box): the default bench frames with the stages one after the other.
# usage: bash scripts/r04_iso_quick.sh OUTDIR TAG
OUT=${1:-gpurun_out/iso}
TAG=${2:-iso}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
RAW=/tmp/iso_raw_$$
mkdir -p $RAW
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/iso -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --e2e-frames 0 --no-libjxl-tables --no-pipeline > $R/$OUT/${TAG}_bench_isolated.json 2> $RAW/iso.log
cp $(find $RAW/iso -name "*kernel_stats.csv" | head -1) $R/$OUT/${TAG}_kernel_stats_isolated.csv
head -12 $R/$OUT/${TAG}_kernel_stats_isolated.csv | cut -c1-220
rm -rf $RAW
