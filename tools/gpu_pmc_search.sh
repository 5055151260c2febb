#!/bin/bash
# HBM bytes of the search kernels (FETCH_SIZE / WRITE_SIZE passes)  ->  gpurun_out/pmc_search/
R=$PWD; O=$R/gpurun_out/pmc_search; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > $O/write.log 2>&1
cd $R
python tools/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1
grep -A3 "sim_\|gemm_tn\|attention_bwd16\|attention_fwd16\|select_radix\|rescore" $O/pmc_summary.txt | head -80
