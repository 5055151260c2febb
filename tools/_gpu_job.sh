set -u
O=gpurun_out/r06_l; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "t5 or T5 or decoder or monot5" > $O/pytest_t5.log 2>&1; echo "t5 rc=$?"
grep -E "passed|failed|Error|^E  |T5 .* float16|decoder position" $O/pytest_t5.log | cut -c1-420 | tail -16
