set -u
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
timeout 600 python tools/gtr_bench.py --dtype float16 2>/dev/null | tail -1 | cut -c1-600
timeout 600 python tools/rerank_bench.py --precision f16 2>/dev/null | tail -1 | cut -c1-600
OM_ENCODER_TWO_PLANE=1 timeout 600 python tools/rerank_bench.py --precision f16 2>/dev/null | tail -1 | cut -c1-200
