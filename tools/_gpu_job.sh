set -u
O=gpurun_out/r06_stagger; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
Q="--no-cpu-baseline --no-extra --no-parity --no-search"
for round in 1 2; do for cfg in "0 2" "16 2" "32 2" "12 4" "48 2"; do set -- $cfg
  OM_GEMM_STAGGER=$1 OM_GEMM_STAGGER_PH=$2 timeout 300 python bench.py --steps 10 --warmup 3 $Q > $O/b_$1_$2_$round.json 2>$O/err.log
  echo "stagger=$1 ph=$2 $(grep -o '"value": [0-9.]*' $O/b_$1_$2_$round.json | head -1) $(grep -o '"frac": [0-9.]*' $O/b_$1_$2_$round.json | head -1)"
done; done
for cfg in "0 2" "32 2"; do set -- $cfg
  OM_GEMM_STAGGER=$1 OM_GEMM_STAGGER_PH=$2 timeout 300 python tools/epilogue_trace.py 2>/dev/null | tail -1 | cut -c1-900
done
