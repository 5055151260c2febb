#!/bin/bash
# training step: weight-gradient lane on / off and workgroup counts of the weight-gradient launches, interleaved  ->  gpurun_out/train_ab/
R=$PWD; O=$R/gpurun_out/train_ab; mkdir -p $O; rm -f $O/*.json
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
for r in 1 2; do
  for w in 0 2 3 4 8; do
    timeout 300 python tools/train_bench.py --steps 30 --wgrad-wgs $w >> $O/train_lane.json 2>$O/train.err
  done
  OM_TRAIN_WGRAD_STREAM=0 timeout 300 python tools/train_bench.py --steps 30 >> $O/train_nolane.json 2>$O/train.err
done
cut -c90-300 $O/train_lane.json; echo nolane; cut -c90-300 $O/train_nolane.json
