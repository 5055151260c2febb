#!/bin/bash
# one counter pass of the final batched weight-gradient kernel (five stages, v_dot2c bias sums)  ->  gpurun_out/r3t/
R=$PWD; O=$R/gpurun_out/r3t; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
cd /tmp; export TMPDIR=/tmp
timeout 45 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- $R/build/selftest tn 9216 0 > $O/p1.log 2>&1
cd $R; python tools/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1; find $O -name "*.csv" -size +1M -delete
grep -A6 "gemm_tn_wide_kernel" $O/pmc_summary.txt | head -40
