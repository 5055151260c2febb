#!/bin/bash
# PMC passes (MFMA busy, waits, LDS) of the batched weight-gradient kernel and of the small-batch scan  ->  gpurun_out/r3p/
R=$PWD; O=$R/gpurun_out/r3p; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
cd /tmp; export TMPDIR=/tmp
for what in tn scan; do
  if [ $what = tn ]; then CMD="$R/build/selftest tn 9216 0"; else CMD="python $R/tools/search_shapes.py --queries 1 64 128"; fi
  mkdir -p $O/$what
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/$what/pmc1 -- $CMD > $O/$what/p1.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/$what/pmc2 -- $CMD > $O/$what/p2.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/$what/pmc3 -- $CMD > $O/$what/p3.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/$what/pmc4 -- $CMD > $O/$what/p4.log 2>&1
  python $R/tools/summarize_pmc.py $O/$what > $O/${what}_pmc_summary.txt 2>&1
  find $O/$what -name "*.csv" -size +2M -delete
done
cd $R
grep -A22 "gemm_tn_wide_kernel" $O/tn_pmc_summary.txt | head -60
grep -A22 "sim_stream_reg" $O/scan_pmc_summary.txt | head -90
