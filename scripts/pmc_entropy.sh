#!/bin/bash
# Stall attribution of the entropy launch (measurement aid): SQ counters of `bench.py --no-pipeline`, one rocprofv3 pass
# per counter set. Raw rocprofv3 output stays in /tmp on the GPU box; only the per-kernel summary goes to OUTDIR.
# usage (on the GPU box): bash scripts/pmc_entropy.sh OUTDIR [extra bench args]
OUT=${1:-gpurun_out/pmc_entropy}
shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
RAW=/tmp/pmc_raw_$$
mkdir -p $RAW
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>&1 | grep -o "SQ_[A-Z_0-9]*" | sort -u > $R/$OUT/counters.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS SQ_IFETCH SQ_WAVES SQ_INSTS_VMEM_WR"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --output-format csv --pmc $P -d $RAW/p$i -o p$i -- python3 $R/bench.py --no-pipeline --steps 2 --warmup 1 --no-cpu-baseline "$@" > $RAW/p$i.log 2>&1 || echo "pass $i failed"
  tail -3 $RAW/p$i.log | cut -c1-400 > $R/$OUT/p$i.tail.txt
  i=$((i+1))
done
python3 - $RAW > $R/$OUT/summary.txt 2>&1 <<'PY'
import csv, glob, collections, sys
for p in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"][:60]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for k in agg:
        if "entropy" in k or "modular_streams" in k or "k_enc_" in k or "filter_rows" in k or "idct_fast<short, 4, 4" in k or "idct_fast<short, 1, 1" in k:
            print(k, {c: round(v[1] / v[0]) for c, v in agg[k].items()}, "dispatches", max(v[0] for v in agg[k].values()))
PY
cat $R/$OUT/summary.txt
rm -rf $RAW
