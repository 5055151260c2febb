#!/bin/bash
# rocprofv3 PMC passes over the native GEMM micro-benchmark (run on the GPU box from the repo root)
R=$PWD; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- $R/build/selftest prof > $R/gpurun_out/pmc_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE
run write WRITE_SIZE
cd $R; ls gpurun_out/pmc_*/*/ 2>/dev/null | head -30
