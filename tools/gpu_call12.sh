#!/bin/bash
R=$PWD; O=$R/gpurun_out/call12; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest full > $O/full.log 2>&1; echo "selftest rc=$?" >> $O/full.log
timeout 600 python -m pytest tests -m gpu -q -k "search or topk or index or retriev or drivers" > $O/pytest_search.log 2>&1; echo "rc=$?" >> $O/pytest_search.log
timeout 600 python tools/search_shapes.py > $O/search_shapes.jsonl 2>$O/search_shapes.err
grep "FAIL\|SELFTEST\|rc=" $O/full.log | tail -8; tail -3 $O/pytest_search.log; cat $O/search_shapes.jsonl
