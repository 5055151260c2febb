#!/bin/bash
R=$PWD; O=$R/gpurun_out/call3; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 400 $R/build/selftest gen7 > $O/gen7.log 2>&1; echo "selftest rc=$?" >> $O/gen7.log
timeout 120 $R/build/selftest quick > $O/quick.log 2>&1; echo "selftest rc=$?" >> $O/quick.log
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err
cat $O/gen7.log; tail -5 $O/quick.log; grep -v "^$" $O/pytest.log | grep "^\[\|passed\|failed\|Error\|error\|rc=" | tail -40; cat $O/bench.json
