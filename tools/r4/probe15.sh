#!/bin/bash
# round 4, probe 15: tile-walk group size of the persistent GEMMs (row tiles whose A panels stay in an XCD's L2) under the power limit
R=$PWD; O=$R/gpurun_out/r4_probe15; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
for round in 1 2; do for g in 8 1 2 4 16 32; do
  OM_GEMM_GROUP_M=$g timeout 300 python bench.py --steps 10 --warmup 3 --no-search --no-cpu-baseline --no-extra --no-parity > $O/bench_g${g}_$round.json 2>$O/bench.err
  python -c "
import json; j=json.load(open('$O/bench_g${g}_$round.json')); print('group_m $g', j['value'], j['roofline']['achieved'])"
done; done
