#!/bin/bash
# round 4, probe 7: gelu'(f) on the training tape (OM_TRAIN_TAPE_GRAD = 0 / 1): training tests, then steps/s interleaved
R=$PWD; O=$R/gpurun_out/r4_probe7; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "train or grad or trainer or cache" > $O/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_train.log
for round in 1 2 3; do for v in 0 1; do
  OM_TRAIN_TAPE_GRAD=$v timeout 300 python tools/train_bench.py --steps 30 >> $O/train_tape$v.json 2>$O/train.err
done; done
for v in 0 1; do echo "OM_TRAIN_TAPE_GRAD=$v"; grep -o '"value": [0-9.]*\|"loss": [0-9.]*' $O/train_tape$v.json | tr '\n' ' '; echo; done
