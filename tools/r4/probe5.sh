#!/bin/bash
# round 4, probe 5: the index scan on the continuous ring (sim_filter_kernel7c) -- search parity tests, then the search leg with OM_GEMM_CONT = 3 / 7
R=$PWD; O=$R/gpurun_out/r4_probe5; mkdir -p $O; rm -f $O/*.log $O/*.json
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "search or topk or retriev or index or flat" > $O/pytest_search.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_search.log
for round in 1 2; do for v in 3 7; do
  OM_GEMM_CONT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/bench_cont${v}_$round.json 2>$O/bench.err
  python - <<PY
import json
j=json.load(open("$O/bench_cont${v}_$round.json"))
s=j["search"]; p=j.get("parity") or {}
print("OM_GEMM_CONT=$v search", s["value"], "q/s scan", s["scan_kernel"], "parity", (p.get("search_full") or {}).get("wrong"), (p.get("search_full") or {}).get("id_sets_identical"), (p.get("search") or {}).get("f16_rescore"))
PY
done; done
