#!/bin/bash
# round 4, probe 9: does the small-batch index pass follow its matrix-core work?  Q = 1 / 64 / 128 with all, half and none of the MFMAs (timing only)
R=$PWD; O=$R/gpurun_out/r4_probe9; mkdir -p $O; rm -f $O/*
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
for round in 1 2; do for dbg in 0 8 16; do
  echo "OM_SEARCH_DEBUG=$dbg" >> $O/small.log
  OM_SEARCH_DEBUG=$dbg timeout 300 python tools/search_shapes.py --queries 1 32 64 128 >> $O/small.log 2>$O/err.log
done; done
grep -o 'OM_SEARCH_DEBUG=[0-9]*\|"queries": [0-9]*\|"ms[a-z_]*": [0-9.]*' $O/small.log | tr '\n' ' ' | sed 's/OM_SEARCH/\nOM_SEARCH/g'
