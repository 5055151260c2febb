set -u
O=gpurun_out/r06_fc; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "long_sequences or beyond or tile_at or packed_rows_beyond or t5_rel or rmsnorm" > $O/pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|Error|^E  " $O/pytest.log | cut -c1-300 | tail -6
for v in 1 5 1 5; do OM_ATTENTION_FAST=$v timeout 300 python tools/long_encode_bench.py 2>/dev/null | tail -1 | cut -c1-400; done
for v in 1 5; do OM_ATTENTION_FAST=$v timeout 300 python tools/train_bench.py --precision f16 --passages 16x512 --steps 20 2>&1 | tail -1 | cut -c90-190; done
