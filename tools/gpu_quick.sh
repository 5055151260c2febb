#!/bin/bash
# quick check of a kernel change: all GPU tests, a short bench line, the per-kernel breakdown  ->  gpurun_out/quick/
R=$PWD; O=$R/gpurun_out/quick; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-extra > $O/bench.json 2>$O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-search --no-cpu-baseline --no-extra --no-parity > $O/prof_stats.log 2>&1
cd $R
tail -3 $O/pytest.log; cat $O/bench.json | cut -c1-600
f=$(ls -t $(find $O/prof_stats -name "*kernel_stats.csv") | head -1); head -8 "$f" | cut -c1-200
