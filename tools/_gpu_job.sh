set -u
O=gpurun_out/r06_j; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 1500 python -m pytest tests/test_gpu_parity_base.py -m gpu -q -s -k "config1 or bit_identical or bert_large or gtr" > $O/pytest_base.log 2>&1; echo "base rc=$?"
grep -E "passed|failed|xfail|Error|against the reference's float16|dMRR" $O/pytest_base.log | cut -c1-300 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "encoder or fused or packed or training_step or train_forward or gemm or residual" > $O/pytest_enc.log 2>&1; echo "enc rc=$?"; grep -E "passed|failed|Error" $O/pytest_enc.log | tail -4
timeout 300 python tools/epilogue_trace.py 2>/dev/null | tail -1 | tee $O/trace.json
for round in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-search --no-cpu-baseline --no-extra --no-parity > $O/bench_$round.json 2>/dev/null
  echo "$(grep -o '"value": [0-9.]*' $O/bench_$round.json | head -1) $(grep -o '"achieved": [0-9.]*' $O/bench_$round.json | head -1)"
done
