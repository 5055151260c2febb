#!/bin/bash
# single-query search: kernel timeline (launch gaps, per-kernel time)  ->  gpurun_out/q1/
R=$PWD; O=$R/gpurun_out/q1; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/search_shapes.py --queries 1 8 64 > $O/shapes.jsonl 2>$O/err.log
cd $R; cat $O/shapes.jsonl
python - <<'PY'
import csv,glob
rows=[]
for p in glob.glob('gpurun_out/q1/trace/**/*kernel_trace.csv', recursive=True):
    rows+=list(csv.DictReader(open(p)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last search of the 1-query batch: find the stream kernels; print the final 3 searches' worth of kernels for nq=1
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'sim_stream_kernel' in n]
print('stream launches',len(idx))
# print a window: all kernels between the first stream kernel of the 2nd search and the next 40 kernels
if idx:
    s=max(0,idx[0]-12); prev=None
    for r in rows[s:s+90]:
        st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        gap=(st-prev)/1e3 if prev else 0
        print(f"{r['Kernel_Name'][:46]:46s} grid={r.get('Grid_Size','?'):>9s} dur={ (en-st)/1e3:9.1f} us gap={gap:7.1f}")
        prev=en
PY
