#!/bin/bash
R=$PWD; O=$R/gpurun_out/call10; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc1 -- $R/build/selftest tn 9216 80 > $O/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc2 -- $R/build/selftest tn 9216 80 > $O/pmc2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM  SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/pmc3 -- $R/build/selftest tn 9216 80 > $O/pmc3.log 2>&1
cd $R
python tools/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt | head -60; tail -3 $O/pmc3.log
