#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
bash tools/gpu_g7probe.sh > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_parity_base.py tests/test_gpu_parity.py -m gpu -q -s -x -k "config1 or bit_identical or fused" > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log; grep -E "config 1|passed|failed|rc=" $O/pytest_b.log | cut -c1-330
python3 tools/cold_probe.py 40 | cut -c1-200
