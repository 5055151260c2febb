#!/bin/bash
# round 4, probe 1: (a) tile phases vs number of active CUs, (b) staggered start, (c) store cache policies, (d) two half batches on two streams
R=$PWD; O=$R/gpurun_out/r4_probe1; mkdir -p $O; rm -f $O/*.log
export TMPDIR=/tmp
rocm-smi --showclocks --showpower > $O/smi_before.txt 2>&1
( G7_MODE=1 timeout 300 build/g7probe_pol1 ) > $O/mode1_pol1.log 2>&1
for round in 1 2; do
  for p in 1 2 3 0; do timeout 120 build/g7probe_pol$p >> $O/policies.log 2>&1; done
done
( G7_MODE=1 timeout 300 build/g7probe_pol2 ) > $O/mode1_pol2.log 2>&1
timeout 600 python tools/two_stream_probe.py --precision f16 > $O/two_stream_f16.log 2>&1
echo "two_stream rc=$?"
tail -40 $O/two_stream_f16.log
grep -v CHECK $O/policies.log | sort | head -80
