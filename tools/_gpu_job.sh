set -u
O=gpurun_out/r06_p; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "few_row or pending or small or bert_tiny or retriever_end_to_end or weight_streaming or layernorm or row_reduction or rmsnorm" > $O/pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|Error|^E  " $O/pytest.log | cut -c1-400 | tail -8
for v in 64 0 64 0; do
OM_FEW_ROWS_LN_FUSE=$v timeout 600 python tools/small_forward_bench.py --limits 1024 --iters 300 --shapes 1x32,2x32,1x64,4x32,1x128 2>/dev/null | tail -1 | cut -c90-400
done
timeout 300 python tools/few_rows_graph_probe.py --shapes 1x32 2>/dev/null | tail -1
