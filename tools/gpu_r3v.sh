#!/bin/bash
# round-3 FINAL validation: the driver's bench command FIRST on the fresh box, the whole GPU test suite, the native self-test, then a
# rocprofv3 kernel-trace summary of the same bench command  ->  gpurun_out/r3x/
R=$PWD; O=$R/gpurun_out/r3x; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | tail -3
timeout 300 $R/build/selftest full > $O/selftest_full.log 2>&1; echo "selftest rc=$?"; tail -2 $O/selftest_full.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/prof_stats.log 2>&1; echo "prof rc=$?"
cd $R; find $O/prof_stats -name "*kernel_stats.csv" | head -2
