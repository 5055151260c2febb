#!/bin/bash
# attention: non-temporal loads / stores -- kernel alone, encoder parity tests, encode bench A/B  ->  gpurun_out/r3j/
R=$PWD; O=$R/gpurun_out/r3j; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 120 $R/build/selftest attn > $O/attn.log 2>&1; grep "non-temporal\|variant=0" $O/attn.log
timeout 600 python -m pytest tests -m gpu -q -x -k "encoder or chain or bit_identical or t5 or attention" > $O/pytest_enc.log 2>&1; echo "rc=$?" >> $O/pytest_enc.log; tail -3 $O/pytest_enc.log
cd /tmp; export TMPDIR=/tmp
for val in 0 1 3 0 1 3; do
  OM_ATTENTION_NT=$val timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-search --no-cpu-baseline --no-extra --no-parity > $O/bench_$val.json 2>$O/bench_$val.err
  echo "OM_ATTENTION_NT=$val $(grep -o '"value": [0-9.]*' $O/bench_$val.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$val.json | head -1)"
done
