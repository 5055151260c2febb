#!/bin/bash
R=$PWD; O=$R/gpurun_out/g7probe; mkdir -p $O; rm -f $O/probe.log
for round in 1 2; do for a in 0 4; do build/g7probe_s1k0_$a >> $O/probe.log 2>&1; build/g7probe_s1k0_$a zero >> $O/probe.log 2>&1; done; done
grep -c ABL $O/probe.log
