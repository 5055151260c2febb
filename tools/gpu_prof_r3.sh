#!/bin/bash
# round-3 profile of bench.py: rocprofv3 --kernel-trace --stats, then the FETCH_SIZE / WRITE_SIZE passes (each alone)
R=$PWD; rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_FETCH_SIZE $R/gpurun_out/prof_WRITE_SIZE
bash tools/prof_round.sh > /dev/null 2>&1
python3 tools/summarize_prof.py r03_bench_v2 r03
tail -2 gpurun_out/prof_stats.log | cut -c1-600 > gpurun_out/prof_stats_line.json
