#!/bin/bash
# last validation of round 3 (generation-6 accumulators pinned in AGPRs): the driver's bench command, the GPU suite, the native self-test  ->  gpurun_out/r3z/
R=$PWD; O=$R/gpurun_out/r3z; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | tail -3
timeout 200 $R/build/selftest full > $O/selftest_full.log 2>&1; echo "selftest rc=$?"; tail -1 $O/selftest_full.log
