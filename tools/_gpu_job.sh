set -u
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
for cfg in 56x144 42x192; do for v in 1 3; do
  echo "t5 $cfg FAST=$v: $(OM_ATTENTION_FAST=$v timeout 300 python tools/train_bench.py --arch t5 --precision f16 --passages $cfg --steps 20 2>&1 | tail -1 | cut -c100-150)"
done; done
