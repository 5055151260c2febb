#!/bin/bash
# small-batch scan: nt vs default cache policy of the index stream, then a kernel timeline of Q = 64 / 128 searches  ->  gpurun_out/r3h/
R=$PWD; O=$R/gpurun_out/r3h; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
timeout 300 python tools/search_shapes.py --queries 1 32 64 128 > $O/shapes_nt.jsonl 2>$O/err.log; echo nt; cut -c1-100 $O/shapes_nt.jsonl
OM_SEARCH_DEBUG=4 timeout 300 python tools/search_shapes.py --queries 1 32 64 128 > $O/shapes_default.jsonl 2>>$O/err.log; echo default; cut -c1-100 $O/shapes_default.jsonl
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/search_shapes.py --queries 64 128 > $O/shapes_prof.jsonl 2>>$O/err.log
cd $R
python - <<'PY'
import csv,glob,collections
rows=[]
for p in glob.glob('gpurun_out/r3h/trace/**/*kernel_trace.csv', recursive=True):
    rows+=list(csv.DictReader(open(p)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
# searches start with init_lists_kernel; take the last search of each NB
starts=[i for i,n in enumerate(names) if 'init_lists_kernel' in n]
def summarize(lo,hi,label):
    agg=collections.OrderedDict(); t0=int(rows[lo]['Start_Timestamp']); t1=int(rows[hi-1]['End_Timestamp']); busy=0
    for r in rows[lo:hi]:
        d=int(r['End_Timestamp'])-int(r['Start_Timestamp']); busy+=d
        k=r['Kernel_Name'][:40]; a=agg.setdefault(k,[0,0]); a[0]+=1; a[1]+=d
    print(label,'span %.1f us, kernels busy %.1f us'%((t1-t0)/1e3,busy/1e3))
    for k,(c,d) in agg.items(): print('   %-40s x%3d  %9.1f us'%(k,c,d/1e3))
segs=[(starts[i],starts[i+1] if i+1<len(starts) else len(rows)) for i in range(len(starts))]
for lo,hi in segs:
    nb=[n for n in names[lo:hi] if 'sim_stream_reg' in n]
    if nb: summarize(lo,hi,nb[0][:40])
PY
