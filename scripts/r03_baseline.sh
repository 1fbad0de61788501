#!/bin/bash
# round-3 baseline: headline line, isolated stage times, the lane kernel's own cycle split
set -e
mkdir -p gpurun_out/r03
python bench.py --steps 5 --warmup 2 --e2e-frames 0 --no-cpu-baseline > gpurun_out/r03/base_bench.json 2> gpurun_out/r03/base_bench.err
cat gpurun_out/r03/base_bench.json
JXLHIP_LANES_PROF=1 python bench.py --no-pipeline --steps 2 --warmup 1 --e2e-frames 0 --no-cpu-baseline > gpurun_out/r03/base_prof.json 2> gpurun_out/r03/base_prof.err
grep "lanes prof" gpurun_out/r03/base_prof.err | tail -3
cat gpurun_out/r03/base_prof.json
