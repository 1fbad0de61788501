#!/bin/bash
# Variants of scripts/ubench_trip3.hip (GPU box): the loop as it is, without its LDS waits, without its store, without both.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -I libjxl_amd/csrc/hip"
$H -o /tmp/t3_base scripts/ubench_trip3.hip 2>/dev/null
$H -DLT_WAITCNT\(N\)='""' -o /tmp/t3_nowait scripts/ubench_trip3.hip 2>/dev/null
$H -DLT_STORE='""' -o /tmp/t3_nostore scripts/ubench_trip3.hip 2>/dev/null
$H -DLT_STORE='""' -DLT_WAITCNT\(N\)='""' -o /tmp/t3_neither scripts/ubench_trip3.hip 2>/dev/null
for v in base nowait nostore neither; do echo "== $v"; timeout -k 5 60 /tmp/t3_$v ${1:-2000}; done
