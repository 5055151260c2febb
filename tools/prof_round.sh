#!/bin/bash
# Round profile of bench.py on the GPU box (run from the repo root):
#   1. kernel-trace stats of the default bench command
#   2. PMC passes (each in its own run, kernel-trace only) for the HBM traffic of the dominant kernel
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $R/gpurun_out/prof_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/prof_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-search --no-extra --no-parity > $R/gpurun_out/prof_$c.log 2>&1
done
cd $R; ls gpurun_out/prof_*/*/ | head -20
