set -u
O=gpurun_out/r06_d; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "packed_rows_training or scaler or float16_training or dropout or trainer_takes or training_step or gradient_cache" > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|packed training step" $O/pytest.log | cut -c1-300 | tail -14
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-search --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('train') or {}; print(json.dumps({k:t.get(k) for k in ('value','loss')})); r=t.get('ragged') or {}; print({k:(r[k]['value'], r[k]['loss']) for k in ('padded','packed') if k in r})" | tee $O/train.txt
