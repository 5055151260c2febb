#!/bin/bash
# bias sums by v_dot2c in the batched weight-gradient kernel: correctness, kernel bench, training  ->  gpurun_out/r3n/
R=$PWD; O=$R/gpurun_out/r3n; mkdir -p $O; rm -f $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 200 $R/build/selftest tn 9216 0 > $O/tn.log 2>&1; echo "rc=$?" >> $O/tn.log; grep "batch\|rc=\|SELFTEST" $O/tn.log
timeout 300 python -m pytest tests -m gpu -q -x -k "weight_gradient or batched or deferred" > $O/pytest_sub.log 2>&1; echo "rc=$?" >> $O/pytest_sub.log; tail -2 $O/pytest_sub.log
for r in 1 2; do
  for b in 0 4; do
    OM_TRAIN_WGRAD_BATCH=$b timeout 200 python tools/train_bench.py --steps 30 2>>$O/train.err | sed "s/^/batch=$b /" >> $O/train.jsonl
  done
done
cut -c1-24,95-160 $O/train.jsonl
