#!/bin/bash
R=$PWD; O=$R/gpurun_out/call2; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest gen7 > $O/gen7.log 2>&1; echo "selftest rc=$?" >> $O/gen7.log
timeout 900 python -m pytest tests/test_gpu_parity_base.py -m gpu -x -q -s > $O/pytest_base.log 2>&1; echo "rc=$?" >> $O/pytest_base.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity_base.py > $O/pytest_rest.log 2>&1; echo "rc=$?" >> $O/pytest_rest.log
cat $O/gen7.log; grep -v "^$" $O/pytest_base.log | tail -40; tail -15 $O/pytest_rest.log
