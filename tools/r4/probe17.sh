#!/bin/bash
# round 4, probes 17 / 18 (the last hours): what a K step of the continuous ring is made of, and the LDS-DMA helper's form
#   build (in the container):
#     hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Iopenmatch_amd/csrc tools/gemm7h_probe.hip -o build/g7h_probe
#     for e in 1 2 4;  do hipcc ... -DG7_DMA_EARLY=$e tools/gemm7h_probe.hip -o build/g7h_probe_e$e; done      # DMA issues front-loaded
#     for a in 0 1 2 3 4 8 12 16 32 35; do hipcc ... -DG7H_ABL=$a tools/gemm7h_probe.hip -o build/g7h_abl_$a; done    # parts compiled out
#     for f in 0 1 2;  do hipcc ... -DG7_DMA_FORM=$f tools/gemm7h_probe.hip -o build/g7h_form_$f; done          # s_nop 4 / s_nop 0 / s_mov_b64 copy
#     for f in 0 2;    do hipcc ... -DG7_DMA_FORM=$f -DVARIANT="\"form$f\"" tools/gemm7_probe.hip -o build/g7probe_form$f; done   # whole launches
#   run (on the GPU box, through gpurun): this script
R=$PWD; O=$R/gpurun_out/r4_g7h; mkdir -p $O
build/g7h_probe > $O/probe.log 2>&1                                                            # 128 x 256 vs 256 x 256 tiles at 9 216 rows
for e in 0 1 2 4; do [ -x build/g7h_probe_e$e ] && { build/g7h_probe_e$e 9216; build/g7h_probe_e$e 131072; }; done > $O/probe_early.log 2>&1
for a in 0 1 2 3 4 8 12 16 32 35; do [ -x build/g7h_abl_$a ] && build/g7h_abl_$a 131072; done > $O/probe_abl.log 2>&1
for r in 1 2; do for f in 0 1 2; do echo "G7_DMA_FORM=$f"; [ -x build/g7h_form_$f ] && build/g7h_form_$f 131072; done; done > $O/probe_form.log 2>&1
for r in 1 2; do for f in 0 2; do [ -x build/g7probe_form$f ] && build/g7probe_form$f; done; done > $O/g7probe_form_ab.log 2>&1
grep -E "G7H_ABL|FORM|x 256 tiles" $O/probe_abl.log $O/probe_form.log | grep -E "ABL=|FORM|qkv|K = 3072" | cut -c1-200
