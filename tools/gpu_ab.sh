#!/bin/bash
# A/B of one environment switch on the encode bench: tools/gpu_ab.sh VAR  ->  gpurun_out/ab/
R=$PWD; O=$R/gpurun_out/ab; mkdir -p $O; V=${1:-OM_ENCODER_PINGPONG}
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for val in 0 1 0 1; do
  env $V=$val timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-search --no-cpu-baseline --no-extra --no-parity > $O/bench_$val.json 2>$O/bench_$val.err
  echo "$V=$val $(cut -c1-260 $O/bench_$val.json | grep -o '"value": [0-9.]*')"
done
for val in 0 1; do
  env $V=$val timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$val -- python $R/bench.py --steps 5 --warmup 2 --no-search --no-cpu-baseline --no-extra --no-parity > $O/prof_$val.log 2>&1
  f=$(find $O/prof_$val -name "*kernel_stats.csv" | head -1); echo "== $V=$val"; head -6 "$f" | cut -d, -f1-4 | cut -c1-60,150-260
done
cd $R; tail -2 $O/pytest.log
