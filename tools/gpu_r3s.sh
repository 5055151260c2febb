#!/bin/bash
# small-batch scan with the two-way unit wait and the incremental issue cursor: correctness, latencies  ->  gpurun_out/r3s/
R=$PWD; O=$R/gpurun_out/r3s; mkdir -p $O; rm -f $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 500 python -m pytest tests -m gpu -q -x -k "search or topk or index or retriev or drivers" > $O/pytest_search.log 2>&1; echo "rc=$?" >> $O/pytest_search.log; tail -2 $O/pytest_search.log
timeout 200 $R/build/selftest full > $O/selftest_full.log 2>&1; echo "selftest rc=$?"; tail -1 $O/selftest_full.log
timeout 300 python tools/search_shapes.py --queries 1 8 32 64 128 256 > $O/shapes.jsonl 2>$O/err.log; cut -c1-100 $O/shapes.jsonl
