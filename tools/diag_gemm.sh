#!/bin/bash
# GEMM diagnostics on the GPU box (run from the repo root): effective shader clock of the real
# kernels, vendor-GEMM reference on the same shapes, memory-path PMC passes of one encoder layer.
R=$PWD; O=$R/gpurun_out/diag; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
for warm in 3 200; do
  for shape in "3072 768" "768 768" "768 3072"; do
    timeout 120 $R/build/selftest trace $shape 131072 0 $warm 2>&1 | grep -v "^blk" >> $O/trace_clock.log
  done
done
timeout 120 $R/build/selftest layer 131072 > $O/layer.log 2>&1
timeout 300 python $R/tools/ref_gemm_torch.py > $O/ref_gemm_torch.jsonl 2>$O/ref_gemm_torch.err
i=0
for c in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_EA0_RDREQ_sum" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" \
         "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc$i -- $R/build/selftest layer 131072 > $O/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $O/pmc_fail.log
done
cd $R
python tools/summarize_pmc.py gpurun_out/diag > gpurun_out/diag/pmc_summary.txt 2>&1
cat gpurun_out/diag/trace_clock.log gpurun_out/diag/layer.log gpurun_out/diag/ref_gemm_torch.jsonl gpurun_out/diag/pmc_summary.txt
