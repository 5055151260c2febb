set -u
O=gpurun_out/r06_dec; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "decoder or monot5 or encoder_decoder" > $O/pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|Error|^E  |T5 decoder position" $O/pytest.log | cut -c1-300 | tail -8
