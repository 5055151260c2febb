#!/bin/bash
# float16 inference mode: parity tests, throughput next to bf16  ->  gpurun_out/f16/
R=$PWD; O=$R/gpurun_out/f16; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests -m gpu -q -s -k "f16 or float16 or bert_large or autocast" > $O/pytest_f16.log 2>&1; echo "rc=$?" >> $O/pytest_f16.log
timeout 600 python bench.py --no-search --no-cpu-baseline > $O/bench_bf16.json 2>$O/bench_bf16.err
timeout 600 python bench.py --precision f16 --no-search --no-cpu-baseline --no-extra --no-parity > $O/bench_f16.json 2>$O/bench_f16.err
grep -v "^$" $O/pytest_f16.log | grep "\[\|passed\|failed\|Error\|rc=" | cut -c1-400 | tail -30
python - <<'PY'
import json
for n in ("bench_bf16","bench_f16"):
    try:
        d=json.loads(open(f"gpurun_out/f16/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["dtype"], json.dumps(d.get("roofline"))[:200]); print(" f16:", json.dumps(d.get("f16"))); print(" parity:", json.dumps((d.get("parity") or {}).get("encode")))
    except Exception as e: print(n, "ERR", e); print(open(f"gpurun_out/f16/{n}.err").read()[-1500:])
PY
