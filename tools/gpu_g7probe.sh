#!/bin/bash
# probe binaries of the generation-7 GEMM tile (tools/gemm7_probe.hip), every binary twice, interleaved
R=$PWD; O=$R/gpurun_out/g7probe; mkdir -p $O; rm -f $O/probe.log
rocm-smi --showclocks --showpower --showperflevel > $O/smi_before.txt 2>&1
for round in 1 2; do
  for b in $(ls build/g7probe_* | sort); do
    timeout 120 $b >> $O/probe.log 2>&1
  done
done
rocm-smi --showclocks --showpower --showperflevel > $O/smi_after.txt 2>&1
grep -c ABL $O/probe.log; grep CHECK $O/probe.log | sort | uniq -c | grep -v " ok " | head
grep -E "sclk|mclk|Power|Perf" $O/smi_before.txt | head; grep -E "sclk|mclk|Power|Perf" $O/smi_after.txt | head
