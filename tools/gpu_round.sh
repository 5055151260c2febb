#!/bin/bash
R=$PWD; O=$R/gpurun_out/round; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 $R/build/selftest full > $O/full.log 2>&1; echo "selftest rc=$?" >> $O/full.log
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2>$O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-parity > $O/prof_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-search --no-cpu-baseline --no-extra --no-parity > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-search --no-cpu-baseline --no-extra --no-parity > $O/pmc_write.log 2>&1
cd $R
python tools/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1
f=$(ls -t $(find $O/prof_stats -name "*kernel_stats.csv") | head -1); head -12 "$f" | cut -c1-150
grep -A3 "gemm_nt_kernel7" $O/pmc_summary.txt | head -40
