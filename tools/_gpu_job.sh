set -u
O=gpurun_out/r06_a; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 1500 python -m pytest tests/test_gpu_parity_base.py -m gpu -q -s -x -k "config1 or bit_identical or bert_large" > $O/pytest_base.log 2>&1; echo "base rc=$?"
grep -E "config 1|bert-large|passed|failed|xfail|Error|error" $O/pytest_base.log | cut -c1-400 | tail -30
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "fold_buffer or encoder_f32_matches or packed_rows_encoder or fused or two_plane or few_rows" > $O/pytest_enc.log 2>&1; echo "enc rc=$?"
grep -E "passed|failed|Error|step/eval" $O/pytest_enc.log | cut -c1-300 | tail -10
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-search > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-1500 $O/bench.json
OM_ENCODER_TWO_PLANE=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-search --no-extra --no-parity > $O/bench_oneplane.json 2>$O/bench1.err; echo "bench(one plane f16) rc=$?"; cut -c1-600 $O/bench_oneplane.json
