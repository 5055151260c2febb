#!/bin/bash
# search round: correctness, shapes, fabric traffic of the scan  ->  gpurun_out/search/
R=$PWD; O=$R/gpurun_out/search; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest full > $O/full.log 2>&1; echo "selftest rc=$?" >> $O/full.log
timeout 600 python -m pytest tests -m gpu -q -k "search or topk or index or retriev or drivers" > $O/pytest_search.log 2>&1; echo "rc=$?" >> $O/pytest_search.log
timeout 600 python tools/search_shapes.py --queries 1 8 32 64 256 1024 6980 > $O/search_shapes.jsonl 2>$O/search_shapes.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/search_shapes.py --queries 6980 > $O/fetch.log 2>&1
cd $R
python tools/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1
grep "FAIL\|SELFTEST\|rc=" $O/full.log | tail -4; tail -2 $O/pytest_search.log; cat $O/search_shapes.jsonl
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for path in glob.glob('gpurun_out/search/pmc_fetch/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'sim_filter_kernel7' in r['Kernel_Name']: agg['k7'].append(float(r['Counter_Value']))
v=agg['k7']; print('sim_filter_kernel7 launches',len(v),'fetch GB total (x2)',2*sum(v)*1024/1e9)
PY
