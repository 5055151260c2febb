#!/bin/bash
# the driver's exact command as the FIRST GPU work on a fresh box, then the per-step series.
R=$PWD; O=$R/gpurun_out/cold; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cold.json 2>$O/bench_cold.err
python3 -c "import json;d=json.load(open('$O/bench_cold.json'));print('cold bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['search']['value'], d['train']['value'])"
sleep 20
python3 tools/cold_probe.py 100 > $O/series_cold.json 2>$O/series_cold.err; cut -c1-1200 $O/series_cold.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/cold_probe.py 30 > $O/trace.log 2>&1
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-parity --no-search > $O/bench_second.json 2>/dev/null
cut -c1-300 $O/bench_second.json
