#!/bin/bash
R=$PWD; O=$R/gpurun_out/call14; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 1200 python -m pytest tests -m gpu -q -k "long_sequences or t5_encoder_decoder or roberta" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | grep "passed\|failed\|FAILED\|rc=" | tail -20
grep -n "^E " $O/pytest.log | head -30
