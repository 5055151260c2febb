#!/bin/bash
# round-3 validation: native self-test, GPU parity tests, smoke, bench (the driver's command)
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest full > $O/full.log 2>&1; echo "selftest rc=$?" >> $O/full.log; tail -2 $O/full.log
timeout 1500 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; cut -c1-400 $O/bench.json
