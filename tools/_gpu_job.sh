set -u
O=gpurun_out/r06_n; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
for tp in 3 0 3 0; do
OM_ENCODER_TWO_PLANE=$tp timeout 600 python tools/small_forward_bench.py --limits 1024 --iters 200 --shapes 1x32,4x32,16x32,1x128,8x128 2>/dev/null | tail -1 | cut -c1-400
done
