#!/bin/bash
R=$PWD; O=$R/gpurun_out/call11; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python -m pytest $R/tests/test_gpu_parity.py -m gpu -q -s -k "kernels_agree and 80" -p no:cacheprovider --rootdir $R > $O/t.log 2>&1
cd $R
tail -5 $O/t.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -i "attention" "$f" | cut -c1-150
