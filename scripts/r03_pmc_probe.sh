#!/bin/bash
# SQ / GRBM counters of the lane kernel alone (scripts/r03_entropy_probe.py, 640 frames, every launch alike): one rocprofv3
# pass per counter set; the per-kernel means go to OUT. usage (GPU box): bash scripts/r03_pmc_probe.sh OUT [BATCH]
OUT=${1:-gpurun_out/r03/pmc_probe}
BATCH=${2:-640}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
RAW=/tmp/pmc_raw_$$
mkdir -p $RAW
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVES"
P3="GRBM_GUI_ACTIVE SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --output-format csv --kernel-trace --kernel-include-regex "k_entropy|k_idct|k_filter|k_dct|k_special|k_color" --pmc $P -d $RAW/p$i -o p$i -- python3 $R/scripts/r03_entropy_probe.py $BATCH ${PROBE_CFG:-base} > $RAW/p$i.log 2>&1 || echo "pass $i failed"
  tail -2 $RAW/p$i.log | cut -c1-300
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
        if "entropy" in k:
            print(k, {c: round(v[1] / v[0]) for c, v in agg[k].items()}, "dispatches", max(v[0] for v in agg[k].values()))
for p in sorted(glob.glob(sys.argv[1] + "/p*/**/*kernel_trace.csv", recursive=True)):
    d = [(r["Kernel_Name"][:50], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(p)) if "entropy" in r["Kernel_Name"]]
    print(p.split("/")[-3], "kernel durations ns:", [x[1] for x in d])
PY
cat $R/$OUT/summary.txt
rm -rf $RAW
