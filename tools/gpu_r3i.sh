#!/bin/bash
# small-batch scan with deferred appends: correctness, latency by batch size, growth of the rounds for 33-128 queries  ->  gpurun_out/r3i/
R=$PWD; O=$R/gpurun_out/r3i; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 500 python -m pytest tests -m gpu -q -x -k "search or topk or index or retriev or drivers" > $O/pytest_search.log 2>&1; echo "rc=$?" >> $O/pytest_search.log; tail -3 $O/pytest_search.log
timeout 300 python tools/search_shapes.py --queries 1 8 32 33 64 96 128 256 > $O/shapes.jsonl 2>$O/err.log; cut -c1-100 $O/shapes.jsonl
for g in 100 200; do echo "growth $g"; OM_SCAN_GROWTH=$g timeout 300 python tools/search_shapes.py --queries 33 64 128 2>>$O/err.log | cut -c1-100; done
