set -u
O=gpurun_out/r06_sw; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmatch_amd/csrc:${LD_LIBRARY_PATH:-}
run() {   # name, env assignments...
  name=$1; shift
  env "$@" timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity_base.py > $O/$name.log 2>&1
  echo "$name : $(grep -E ' passed| failed' $O/$name.log | tail -1) $(grep -E '^FAILED|^E  ' $O/$name.log | head -2 | cut -c1-200 | tr '\n' ' ')"
}
run few_rows_ln_fuse_0 OM_FEW_ROWS_LN_FUSE=0
run skinny_m_0 OM_GEMM_SKINNY_M=0
run two_plane_0 OM_ENCODER_TWO_PLANE=0
run two_plane_7 OM_ENCODER_TWO_PLANE=7
run gemm_cont_1519 OM_GEMM_CONT=1519
run attention_fast_0 OM_ATTENTION_FAST=0
run attention_fast_5 OM_ATTENTION_FAST=5
run train_f16_0 OM_TRAIN_F16=0
