#!/bin/bash
# selftest + GPU tests after the generation-4 removal, then the training A/B
R=$PWD; O=$R/gpurun_out/r3d; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest full > $O/full.log 2>&1; echo "selftest rc=$?" >> $O/full.log; tail -2 $O/full.log
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed" $O/pytest.log
bash tools/gpu_train_ab.sh
