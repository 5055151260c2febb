#!/bin/bash
# round 4, probe 11: store policy on the continuous-ring kernels (plain vs non-temporal), then the evidence passes of the encode leg:
# kernel-trace stats, HBM traffic (FETCH_SIZE / WRITE_SIZE), matrix-core busy and wait counters
R=$PWD; O=$R/gpurun_out/r4_probe11; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
for round in 1 2; do for p in 0 1; do timeout 200 build/g7probe_pol$p > $O/pol${p}_$round.log 2>&1; done; done
for p in 0 1; do echo "policy $p"; grep "cont=3 " $O/pol${p}_2.log | cut -c1-170; done
cd /tmp
Q="--no-cpu-baseline --no-extra --no-parity --no-search"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 $Q > $O/prof.log 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc$i -- python $R/bench.py --steps 3 --warmup 1 $Q > $O/pmc$i.log 2>&1 || echo "pass $i failed"
done
cd $R; f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; head -12 "$f" | cut -c1-200
python tools/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1; grep -A10 "gemm_nt_kernel7[cr]" $O/pmc_summary.txt | head -90
