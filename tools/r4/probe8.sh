#!/bin/bash
# round 4, probe 8: per-kernel times of the training step (rocprofv3 --kernel-trace --stats of tools/train_bench.py)
R=$PWD; O=$R/gpurun_out/r4_probe8; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/train_bench.py --steps 20 > $O/prof.log 2>&1
cd $R; f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_kernel_stats.csv; head -32 "$f" | cut -c1-200
