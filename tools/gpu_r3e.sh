#!/bin/bash
# round 3, late: scan filter rewrite + batched weight gradients -- correctness first, then timings  ->  gpurun_out/r3e/
R=$PWD; O=$R/gpurun_out/r3e; mkdir -p $O; rm -f $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 200 $R/build/selftest tn > $O/tn.log 2>&1; echo "rc=$?" >> $O/tn.log; grep "batch\|rc=\|SELFTEST" $O/tn.log
timeout 500 python -m pytest tests -m gpu -q -x -k "search or topk or index or retriev or drivers or weight_gradient or batched or training or gradient" > $O/pytest_sub.log 2>&1; echo "rc=$?" >> $O/pytest_sub.log; tail -4 $O/pytest_sub.log
timeout 300 python tools/search_shapes.py --queries 1 64 256 1024 6980 > $O/shapes.jsonl 2>$O/shapes.err; cut -c1-200 $O/shapes.jsonl
for r in 1 2; do
  for b in 0 4 12 2; do
    OM_TRAIN_WGRAD_BATCH=$b timeout 200 python tools/train_bench.py --steps 30 2>>$O/train.err | sed "s/^/batch=$b /" >> $O/train.jsonl
  done
done
cut -c1-20,95-200 $O/train.jsonl
timeout 120 $R/build/selftest scantrace > $O/scantrace.log 2>&1; tail -4 $O/scantrace.log
