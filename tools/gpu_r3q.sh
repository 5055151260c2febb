#!/bin/bash
# secondary configurations re-measured at the end of round 3 (GTR-base T5 encode, bert-large cross-encoder), the smoke entry point,
# and the growth of the scan rounds at Q = 6980 after the filter rewrite  ->  gpurun_out/r3q/
R=$PWD; O=$R/gpurun_out/r3q; mkdir -p $O; rm -f $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 200 python tools/gtr_bench.py > $O/gtr.json 2>$O/gtr.err; cut -c1-300 $O/gtr.json
timeout 200 python tools/rerank_bench.py > $O/rerank.json 2>$O/rerank.err; cut -c1-300 $O/rerank.json
for g in 60 40 80 100 60; do echo "growth $g"; OM_SCAN_GROWTH=$g timeout 200 python tools/search_shapes.py --queries 6980 2>>$O/err.log | cut -c1-100 | tee -a $O/growth_$g.jsonl; done
