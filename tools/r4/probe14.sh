#!/bin/bash
# round 4, probe 14: kernels 7c16 / 7r16 (16 x 16 x 32 MFMAs) -- self-test and GPU suite under OM_GEMM_CONT=15, traces and launch times, encode leg A/B
R=$PWD; O=$R/gpurun_out/r4_probe14; mkdir -p $O; rm -f $O/*.log $O/*.json
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
OM_GEMM_CONT=15 timeout 600 build/selftest gen7 4096 > $O/selftest_gen7.log 2>&1; echo "selftest gen7 rc=$?"; grep -c "\[ OK \]" $O/selftest_gen7.log; grep "FAIL" $O/selftest_gen7.log | head
timeout 300 build/g7probe_v15 > $O/g7probe.log 2>&1; echo "probe rc=$?"; grep CHECK $O/g7probe.log | grep -v " ok " | head
grep "cont=[0-9]* " $O/g7probe.log | cut -c1-170 | tail -27
grep -A2 "cont=11 " $O/g7probe.log | grep "trace\|residual" | cut -c1-300 | tail -12
for round in 1 2; do for v in 7 15; do
  OM_GEMM_CONT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-search --no-cpu-baseline --no-extra > $O/bench_cont${v}_$round.json 2>$O/bench.err
  python -c "
import json; j=json.load(open('$O/bench_cont${v}_$round.json')); e=(j.get('parity') or {}).get('encode') or {}; print('OM_GEMM_CONT=$v', j['value'], j['roofline']['achieved'], 'f16 rel ddot', e.get('f16_max_rel_ddot'), 'min cos', e.get('f16_min_cosine'))"
done; done
OM_GEMM_CONT=15 timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E " passed| failed| error" $O/pytest.log | tail -3
