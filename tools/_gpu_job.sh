set -u
O=gpurun_out/r06_s; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 900 python -m pytest tests -m gpu -q -x -k "search or flat_ip or topk or retriev or scan or index" > $O/pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|Error|^E  " $O/pytest.log | cut -c1-300 | tail -4
