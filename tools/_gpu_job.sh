set -u
O=gpurun_out/r06_f; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "residual_stream_variants" > $O/pytest_rs.log 2>&1; echo "rs rc=$?"
grep -E "passed|failed|Error|residual stream" $O/pytest_rs.log | cut -c1-300 | tail -8
timeout 1700 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E " passed| failed| error|^FAILED|^E  " $O/pytest.log | tail -10
