set -u
O=$PWD/gpurun_out/r06_lp; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/tools/train_bench.py --precision f16 --passages 16x512 --steps 10 > $O/prof.log 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
head -12 $O/kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
