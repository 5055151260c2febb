#!/bin/bash
# search change check: native self-test (incl. adversarial row orders), search tests, latency by batch size -> gpurun_out/sq/
R=$PWD; O=$R/gpurun_out/sq; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest full > $O/full.log 2>&1; echo "selftest rc=$?" >> $O/full.log
timeout 600 python -m pytest tests -m gpu -q -k "search or topk or index or retriev or drivers" > $O/pytest_search.log 2>&1; echo "rc=$?" >> $O/pytest_search.log
timeout 600 python tools/search_shapes.py --queries 1 64 256 1024 6980 > $O/shapes.jsonl 2>$O/err.log
grep "FAIL\|SELFTEST\|rc=" $O/full.log | tail -4; tail -2 $O/pytest_search.log; cut -c1-230 $O/shapes.jsonl
