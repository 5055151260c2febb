#!/bin/bash
# One parameterised driver for everything that runs on the GPU box (from the repo root, through gpurun):
#
#   tools/gpu.sh validate [tag]              the driver's bench command FIRST (fresh box), then the GPU suite, smoke, native self-test
#   tools/gpu.sh prof [tag] [bench args]     rocprofv3 --kernel-trace --stats of bench.py (default: the encode + search legs)
#   tools/gpu.sh pmc [tag] SET [bench args]  counter passes, each in its own kernel-trace-only run; SET = hbm | sq | lds | mem
#   tools/gpu.sh ab [tag] VAR v1 v2 ...      interleaved A/B of one environment switch on the encode leg (two rounds)
#   tools/gpu.sh probes [tag] GLOB [env...]  every probe binary matching build/GLOB twice, interleaved (tools/*_probe.hip builds)
#   tools/gpu.sh py [tag] script.py [args]   one python tool (tools/*.py) with its output kept
#   tools/gpu.sh train [tag] [CFG...]        the training step: interleaved A/B of "OM_GEMM_CONT optimizer" pairs (default: "47 fused"
#                                            "15 fused" "47 torch"), then rocprofv3 kernel stats of the default configuration
#   tools/gpu.sh suite [tag] [pytest args]   the GPU suite alone (-m gpu), log kept
#   tools/gpu.sh sweep [tag] VAR v1 v2 .. -- CMD   CMD once per value of the environment variable VAR, twice, interleaved; last JSON line of each kept
#
# Output goes to gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/.
# Rounds 1-3 used one script per experiment (tools/gpu_r3*.sh ...); they are in the git history up to 9937a22; round 4's
# tools/r4/probe*.sh up to 6ae6e48.  New experiments become sub-commands here.
set -u
R=$PWD; sub=${1:-validate}; tag=${2:-$sub}; shift; shift
O=$R/gpurun_out/$tag; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
export TMPDIR=/tmp
QUIET="--no-cpu-baseline --no-extra --no-parity"

case $sub in
validate)
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
  timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; grep -E " passed| failed| error" $O/pytest.log | tail -3
  timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
  [ -x build/selftest ] && { timeout 300 build/selftest full > $O/selftest_full.log 2>&1; echo "selftest rc=$?"; tail -1 $O/selftest_full.log; }
  ;;
prof)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity "$@" > $O/prof.log 2>&1
  cd $R; f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" $O/kernel_stats.csv; head -14 "$f" | cut -c1-170; }
  ;;
pmc)
  set_=${1:-hbm}; shift
  case $set_ in
    hbm) passes=("FETCH_SIZE" "WRITE_SIZE") ;;
    sq)  passes=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS") ;;
    lds) passes=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU") ;;
    mem) passes=("TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_EA0_RDREQ_sum") ;;
    *) echo "unknown counter set $set_"; exit 2 ;;
  esac
  cd /tmp; i=0
  for c in "${passes[@]}"; do
    i=$((i+1))      # counters in their own run, kernel trace only (never with --sys-trace / hip / hsa / memory-copy domains)
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc$i -- python $R/bench.py --steps 3 --warmup 1 $QUIET "$@" > $O/pmc$i.log 2>&1 || echo "pass $i failed"
  done
  cd $R; python tools/summarize_pmc.py $O > $O/pmc_summary.txt 2>&1; grep -A6 "gemm_nt_kernel7\|sim_filter_kernel7\|attention_fwd16" $O/pmc_summary.txt | head -80
  ;;
ab)
  V=$1; shift
  for round in 1 2; do for val in "$@"; do
    env $V=$val timeout 300 python bench.py --steps 10 --warmup 3 --no-search $QUIET > $O/bench_${val}_$round.json 2>$O/bench_${val}_$round.err
    echo "$V=$val $(grep -o '"value": [0-9.]*' $O/bench_${val}_$round.json | head -1) $(grep -o '"achieved": [0-9.]*' $O/bench_${val}_$round.json | head -1)"
  done; done
  ;;
probes)
  G=$1; shift; rm -f $O/probes.log
  for round in 1 2; do for b in $(ls build/$G | sort); do env "$@" timeout 300 $b >> $O/probes.log 2>&1; done; done
  grep -c "TFLOP" $O/probes.log; grep CHECK $O/probes.log | sort | uniq -c | grep -v " ok " | head
  ;;
py)
  s=$1; shift
  timeout 900 python $s "$@" > $O/$(basename $s .py).log 2>&1; echo "rc=$?"; grep -v Warning $O/$(basename $s .py).log | tail -40
  ;;
train)
  [ $# -eq 0 ] && set -- "47 fused" "15 fused" "47 torch"
  cfgs=("$@")
  for round in 1 2; do for cfg in "${cfgs[@]}"; do
    set -- $cfg; c=$1; o=$2; h=${3:-dev}          # "OM_GEMM_CONT optimizer [host]": host = the batch stays in pageable host memory (a copy + sync per step)
    hb=0; [ "$h" = host ] && hb=1
    TRAIN_BENCH_HOST_BATCH=$hb OM_GEMM_CONT=$c timeout 300 python tools/train_bench.py --steps 30 --optimizer $o > $O/train_${c}_${o}_${h}_$round.json 2>$O/train.err
    echo "OM_GEMM_CONT=$c optimizer=$o batch=$h $(grep -o '"steps_per_s": [0-9.]*' $O/train_${c}_${o}_${h}_$round.json) $(grep -o '"loss": [0-9.]*' $O/train_${c}_${o}_${h}_$round.json)"
  done; done
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/train_bench.py --steps 20 > $O/prof.log 2>&1
  cd $R; f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" $O/train_kernel_stats.csv; head -16 "$f" | cut -c1-150; }
  ;;
sweep)
  V=$1; shift; vals=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
  shift
  for round in 1 2; do for val in "${vals[@]}"; do
    env $V=$val timeout 600 "$@" > $O/sweep_${V}_${val}_$round.log 2>$O/sweep.err
    echo "$V=$val $(grep '^{' $O/sweep_${V}_${val}_$round.log | tail -1 | cut -c1-260)"
  done; done
  ;;
suite)
  timeout 1500 python -m pytest tests -m gpu -q -s "$@" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  grep -E " passed| failed| error|^FAILED|^E  " $O/pytest.log | tail -15
  ;;
*) echo "unknown subcommand $sub"; exit 2 ;;
esac
