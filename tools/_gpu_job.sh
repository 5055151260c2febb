set -u
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
for cfg in 64x128 32x256 16x512 24x384; do
  timeout 300 python tools/train_bench.py --precision f16 --passages $cfg --steps 20 2>&1 | tail -1 | cut -c1-260
done
