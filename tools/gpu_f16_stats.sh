#!/bin/bash
# per-kernel breakdown of the float16 encode leg  ->  gpurun_out/f16/prof_stats
R=$PWD; O=$R/gpurun_out/f16; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --precision f16 --steps 5 --warmup 2 --no-search --no-cpu-baseline --no-extra --no-parity > $O/prof_stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_bf16 -- python $R/bench.py --steps 5 --warmup 2 --no-search --no-cpu-baseline --no-extra --no-parity > $O/prof_stats_bf16.log 2>&1
cd $R
for d in prof_stats prof_stats_bf16; do f=$(ls -t $(find $O/$d -name "*kernel_stats.csv") | head -1); echo "== $d"; head -7 "$f" | cut -d, -f1-4 | cut -c1-70,150-230; done
