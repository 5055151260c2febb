#!/bin/bash
# training step with the weight-gradient lane on / off, interleaved  ->  gpurun_out/train_ab/
R=$PWD; O=$R/gpurun_out/train_ab; mkdir -p $O; rm -f $O/*.json
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "weight_gradient or one_pass or training or train or drtrainer or rr or attention_backward or dropout or t5 or roberta or allreduce" > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log
tail -3 $O/pytest_train.log
for r in 1 2; do for v in 1 0; do
  OM_TRAIN_WGRAD_STREAM=$v timeout 300 python tools/train_bench.py --steps 30 >> $O/train_$v.json 2>$O/train_$v.err
done; done
for v in 1 0; do echo "lane=$v"; cat $O/train_$v.json | cut -c1-260; done
