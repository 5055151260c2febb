#!/bin/bash
R=$PWD; O=$R/gpurun_out/q1; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest full > $O/full.log 2>&1; echo "selftest rc=$?" >> $O/full.log
timeout 600 python -m pytest tests -m gpu -q -k "search or topk or index or retriev or drivers" > $O/pytest_search.log 2>&1; echo "rc=$?" >> $O/pytest_search.log
timeout 600 python tools/search_shapes.py --queries 1 8 32 64 6980 > $O/shapes_plain.jsonl 2>$O/err_plain.log
grep "FAIL\|SELFTEST\|rc=" $O/full.log | tail -4; tail -2 $O/pytest_search.log; cut -c1-200 $O/shapes_plain.jsonl
bash tools/gpu_q1.sh | tail -45 | cut -c1-120
