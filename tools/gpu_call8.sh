#!/bin/bash
R=$PWD; O=$R/gpurun_out/call8; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk|power|mclk" | head -6
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "weight_gradient or one_pass or training or train or drtrainer or rr" > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log
timeout 300 python tools/train_bench.py --steps 20 > $O/train.json 2>$O/train.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/tools/train_bench.py --steps 10 > $O/prof_train.log 2>&1
cd $R; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
grep -v "^$" $O/pytest_train.log | grep "^\[\|passed\|failed\|Error\|error\|rc=\|FAIL\|worst" | tail -20; cat $O/train.json; tail -3 $O/train.err
f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); head -32 "$f" | cut -c1-170
