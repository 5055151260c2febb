#!/bin/bash
R=$PWD; O=$R/gpurun_out/call5; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest full > $O/full.log 2>&1; echo "selftest rc=$?" >> $O/full.log
timeout 300 $R/build/selftest gen7 > $O/gen7.log 2>&1; echo "selftest rc=$?" >> $O/gen7.log
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2>$O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-parity > $O/prof_stats.log 2>&1
cd $R
grep "FAIL\|SELFTEST\|BENCH\|rc=" $O/full.log | tail -12; grep "BENCH\|avg over\|SELFTEST\|^--" $O/gen7.log | tail -30; grep -v "^$" $O/pytest.log | grep "^\[\|passed\|failed\|Error\|error\|rc=\|FAIL" | tail -20; cat $O/bench.json
f=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); head -16 "$f" | cut -c1-150
