#!/bin/bash
# round 4, probe 12: attention with masked key tiles skipped -- tests, then the encode leg
R=$PWD; O=$R/gpurun_out/r4_probe12; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E " passed| failed| error" $O/pytest.log | tail -3
for round in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-search --no-cpu-baseline --no-extra > $O/bench_$round.json 2>$O/bench.err
  python -c "
import json; j=json.load(open('$O/bench_$round.json')); print(j['value'], j['roofline']['achieved'], (j.get('parity') or {}).get('encode'))"
done
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-search --no-cpu-baseline --no-extra --no-parity > $O/prof.log 2>&1
cd $R; f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; head -7 "$f" | cut -c1-160
