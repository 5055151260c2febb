#!/bin/bash
# round 4, probe 3: what a global store costs the issuing wave (and whether VALU work hides it); per-iteration stamps of the residual epilogue
R=$PWD; O=$R/gpurun_out/r4_probe3; mkdir -p $O; rm -f $O/*.log
timeout 300 build/store_issue_probe > $O/store_issue.log 2>&1; echo "rc=$?"; cat $O/store_issue.log
timeout 300 build/g7probe_v11 > $O/g7probe.log 2>&1; echo "probe rc=$?"; grep -A2 "resid" $O/g7probe.log | grep -v "^--" | cut -c1-330 | tail -30
