#!/bin/bash
# shader clock under load per GEMM launch, bfloat16 vs float16: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / kernel duration
R=$PWD; O=$R/gpurun_out/clock; mkdir -p $O
export LD_LIBRARY_PATH=$R/openmatch_amd/csrc:$LD_LIBRARY_PATH
cd /tmp; export TMPDIR=/tmp
for p in bf16 f16; do
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/$p -- python $R/bench.py --precision $p --steps 3 --warmup 1 --no-search --no-cpu-baseline --no-extra --no-parity > $O/$p.log 2>&1
done
cd $R
python - <<'PY'
import csv,glob,collections
for p in ("bf16","f16"):
    cnt=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/clock/{p}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            if "gemm_nt_kernel7" in k or "attention_fwd16" in k:
                cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"]=="GRBM_GUI_ACTIVE": dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
    for k in sorted(cnt):
        g=sum(cnt[k]["GRBM_GUI_ACTIVE"])/len(cnt[k]["GRBM_GUI_ACTIVE"]); d=sum(dur[k])/len(dur[k]); m=sum(cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"])/max(1,len(cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"]))
        print(f"{p:5s} {k:60s} n={len(dur[k]):4d} dur={d/1e3:7.1f} us  GUI_ACTIVE/8={g/8:10.0f}  clock={g/8/d:5.2f} GHz  mfma_busy={m/(1024*g/8):.3f}")
PY
