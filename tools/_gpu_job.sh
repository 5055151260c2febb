set -u
O=gpurun_out/r06_l; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "tile_at_a_time or beyond_256" > $O/pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|Error|^E  |tile-at-a-time|training at L" $O/pytest.log | cut -c1-330 | tail -40
