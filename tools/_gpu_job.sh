set -u
O=gpurun_out/r06_pk; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "packed or beyond or tile_at" > $O/pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|Error|^E  " $O/pytest.log | cut -c1-300 | tail -8
