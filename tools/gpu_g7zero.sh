#!/bin/bash
# K loop on random vs zero-filled operands (the power limit), and the eight-wave form of the loop
R=$PWD; O=$R/gpurun_out/g7probe; mkdir -p $O; rm -f $O/probe.log
for round in 1 2; do
  build/g7probe_final_0 >> $O/probe.log 2>&1; build/g7probe_final_0 zero >> $O/probe.log 2>&1
  build/g7probe_final_4 >> $O/probe.log 2>&1; build/g7probe_final_4 zero >> $O/probe.log 2>&1
  build/g8probe_p0 >> $O/probe.log 2>&1
done
grep -c ABL $O/probe.log
