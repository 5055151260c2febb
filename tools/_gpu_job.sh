set -u
O=gpurun_out/r06_m; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_base.py -m gpu -q -x -k "train or attention or packed or t5 or gradient or grad" > $O/pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|Error|^E  " $O/pytest.log | cut -c1-300 | tail -4
for cfg in 56x162 40x208; do echo "$cfg $(timeout 300 python tools/train_bench.py --precision f16 --passages $cfg --steps 20 2>&1 | tail -1 | cut -c90-150)"; done
