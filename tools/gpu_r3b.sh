#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_base.py tests/test_gpu_parity.py -m gpu -q -s -x -k "config1 or bit_identical or fused or search_properties or retriever_end_to_end" > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log; grep -E "config 1|passed|failed|identical|rc=" $O/pytest_b.log | cut -c1-400
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/cold_probe.py 24 > $O/trace.log 2>&1
cd $R
OM_ENCODER_TWO_PLANE=0 python3 tools/cold_probe.py 30 | cut -c1-200
python3 tools/cold_probe.py 30 | cut -c1-200
