#!/bin/bash
# round 4, probe 10: small-batch scan with the next unit's first fragments read a unit early -- parity tests, then latencies
R=$PWD; O=$R/gpurun_out/r4_probe10; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_logic.py -q -x -k "search or topk or retriev or index or flat or append or deferred" > $O/pytest_search.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_search.log
for round in 1 2 3; do timeout 300 python tools/search_shapes.py --queries 1 8 32 64 128 >> $O/small.log 2>$O/err.log; done
grep -o '"queries": [0-9]*\|"ms[a-z_]*": [0-9.]*' $O/small.log | tr '\n' ' ' | sed 's/"queries": 1 /\n"queries": 1 /g'
