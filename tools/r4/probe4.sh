#!/bin/bash
# round 4, probe 4: the residual variants on the continuous ring (kernel 7r) -- self-test, traces and launch times, encode leg A/B, GPU suite
R=$PWD; O=$R/gpurun_out/r4_probe4; mkdir -p $O; rm -f $O/*.log $O/*.json
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 600 build/selftest gen7 4096 > $O/selftest_gen7.log 2>&1; echo "selftest gen7 rc=$?"; grep -c "\[ OK \]" $O/selftest_gen7.log; grep "FAIL" $O/selftest_gen7.log | head
timeout 300 build/g7probe_v12 > $O/g7probe.log 2>&1; echo "probe rc=$?"; grep CHECK $O/g7probe.log | grep -v " ok " | head
grep -A2 "cont=[03] " $O/g7probe.log | grep -v "^--" | cut -c1-330 | tail -64
for round in 1 2; do for v in 0 1 3; do
  OM_GEMM_CONT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-search --no-cpu-baseline --no-extra --no-parity > $O/bench_cont${v}_$round.json 2>$O/bench.err
  echo "OM_GEMM_CONT=$v $(grep -o '"value": [0-9.]*' $O/bench_cont${v}_$round.json | head -1) $(grep -o '"achieved": [0-9.]*' $O/bench_cont${v}_$round.json | head -1)"
done; done
timeout 1500 python -m pytest tests -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E " passed| failed| error" $O/pytest.log | tail -3; grep "config 1 spread" $O/pytest.log | cut -c1-600
