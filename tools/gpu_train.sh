#!/bin/bash
# training-step round: gradient tests, steps/s, kernel breakdown  ->  gpurun_out/train/
R=$PWD; O=$R/gpurun_out/train; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "weight_gradient or one_pass or training or train or drtrainer or rr or attention_backward or dropout or t5 or roberta" > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log
timeout 300 python tools/train_bench.py --steps 20 > $O/train.json 2>$O/train.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/tools/train_bench.py --steps 10 > $O/prof_train.log 2>&1
cd $R
tail -3 $O/pytest_train.log; cat $O/train.json
