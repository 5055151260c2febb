set -u
O=gpurun_out/r06_t; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
for r in 1 2; do for p in 0 1; do
  timeout 300 python tools/train_bench.py --arch t5 --precision f16 --ragged --packed $p --steps 20 2>&1 | tail -1 | cut -c1-300
done; done
timeout 300 python tools/train_bench.py --arch t5 --precision f16 --steps 20 2>&1 | tail -1 | cut -c1-300
