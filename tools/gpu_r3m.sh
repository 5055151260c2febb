#!/bin/bash
# five-stage batched weight-gradient kernel + AGPR-pinned small-batch scan: correctness, kernel bench, latencies, training  ->  gpurun_out/r3m/
R=$PWD; O=$R/gpurun_out/r3m; mkdir -p $O; rm -f $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 200 $R/build/selftest tn 9216 0 > $O/tn.log 2>&1; echo "rc=$?" >> $O/tn.log; grep "batch\|rc=\|SELFTEST" $O/tn.log
timeout 500 python -m pytest tests -m gpu -q -x -k "search or topk or index or retriev or drivers or weight_gradient or batched or deferred" > $O/pytest_sub.log 2>&1; echo "rc=$?" >> $O/pytest_sub.log; tail -3 $O/pytest_sub.log
timeout 300 python tools/search_shapes.py --queries 1 8 32 64 128 256 > $O/shapes.jsonl 2>$O/err.log; cut -c1-100 $O/shapes.jsonl
for r in 1 2; do
  for b in 0 4 12; do
    OM_TRAIN_WGRAD_BATCH=$b timeout 200 python tools/train_bench.py --steps 30 2>>$O/train.err | sed "s/^/batch=$b /" >> $O/train.jsonl
  done
done
cut -c1-24,95-160 $O/train.jsonl
