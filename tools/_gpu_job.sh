set -u
O=gpurun_out/r06_t5b; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_base.py -m gpu -q -x -k "t5 or tile_at_a_time or attention_backward or beyond_256" > $O/pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|Error|^E  " $O/pytest.log | cut -c1-330 | tail -6
for r in 1 2; do
timeout 300 python tools/train_bench.py --arch t5 --precision f16 --steps 20 2>&1 | tail -1 | cut -c90-200
OM_ATTENTION_FAST=0 timeout 300 python tools/train_bench.py --arch t5 --precision f16 --steps 20 2>&1 | tail -1 | cut -c90-200
done
timeout 300 python tools/train_bench.py --arch t5 --precision f16 --ragged --packed 1 --steps 20 2>&1 | tail -1 | cut -c90-200
