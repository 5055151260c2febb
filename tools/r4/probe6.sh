#!/bin/bash
# round 4, probe 6: fabric traffic of one 6980-query search (FETCH_SIZE / WRITE_SIZE passes) and the per-XCD query-group size
R=$PWD; O=$R/gpurun_out/r4_probe6; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
for g in 8 4 6 12 16; do
  OM_SCAN_QGROUP=$g timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-parity > $O/bench_qg$g.json 2>$O/bench.err
  python -c "
import json; j=json.load(open('$O/bench_qg$g.json')); s=j['search']; print('qgroup $g', s['value'], 'q/s', s['scan_kernel'])"
done
cd /tmp
for g in 8 4; do
  OM_SCAN_QGROUP=$g timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/qg$g/pmc1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-parity > $O/pmc_qg$g.log 2>&1
  python $R/tools/summarize_pmc.py $O/qg$g > $O/pmc_summary_qg$g.txt 2>&1; grep -A2 "sim_filter_kernel7c" $O/pmc_summary_qg$g.txt | head -8
done
