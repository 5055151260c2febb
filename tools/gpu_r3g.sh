#!/bin/bash
# register-resident small-batch scan (Q <= 128): correctness, then latency by batch size, new vs round-2 kernels  ->  gpurun_out/r3g/
R=$PWD; O=$R/gpurun_out/r3g; mkdir -p $O; rm -f $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 500 python -m pytest tests -m gpu -q -x -k "search or topk or index or retriev or drivers" > $O/pytest_search.log 2>&1; echo "rc=$?" >> $O/pytest_search.log; tail -3 $O/pytest_search.log
timeout 300 python tools/search_shapes.py --queries 1 8 32 33 64 96 128 129 256 > $O/shapes_new.jsonl 2>$O/err.log; cut -c1-120 $O/shapes_new.jsonl
OM_SEARCH_DEBUG=2 timeout 300 python tools/search_shapes.py --queries 1 8 32 33 64 96 128 > $O/shapes_old.jsonl 2>>$O/err.log; echo old; cut -c1-120 $O/shapes_old.jsonl
