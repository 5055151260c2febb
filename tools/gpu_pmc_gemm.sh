#!/bin/bash
# MFMA-busy / LDS / wait counters of the encoder kernels (separate --pmc passes)  ->  gpurun_out/pmc_gemm/
R=$PWD; O=$R/gpurun_out/pmc_gemm; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
CMD="python $R/bench.py --steps 2 --warmup 1 --no-search --no-cpu-baseline --no-extra --no-parity"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc1 -- $CMD > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc2 -- $CMD > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc3 -- $CMD > $O/p3.log 2>&1
cd $R
python tools/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1
grep -A13 "gemm_nt_kernel7\|attention_fwd16" $O/pmc_summary.txt | head -120
