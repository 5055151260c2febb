#!/bin/bash
R=$PWD; O=$R/gpurun_out/call7; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest full > $O/full.log 2>&1; echo "selftest rc=$?" >> $O/full.log
timeout 400 $R/build/selftest scantrace 8841823 6980 200 100 60 40 25 15 > $O/scantrace.log 2>&1; echo "rc=$?" >> $O/scantrace.log
timeout 300 $R/build/selftest gen7 > $O/gen7.log 2>&1; echo "selftest rc=$?" >> $O/gen7.log
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
grep "FAIL\|SELFTEST\|rc=" $O/full.log | tail -12; cat $O/scantrace.log; grep "BENCH\|avg over\|SELFTEST\|^--" $O/gen7.log | tail -30; grep -v "^$" $O/pytest.log | grep "^\[\|passed\|failed\|Error\|error\|rc=\|FAIL" | tail -20; cat $O/bench.json
