#!/usr/bin/env python
"""Average kernel duration per (kernel, grid) from a rocprofv3 kernel trace of tools/skinny_sweep.py:
python tools/skinny_trace_summary.py <kernel_trace.csv>  ->  one line per (N, K is implied by the launch order, M rows, config)."""
import csv, re, sys, collections
acc = collections.OrderedDict()
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"]
    if "skinny" not in name and "gemm_nt_kernel" not in name:
        continue
    m = re.search(r"skinny_kernelI\w+?Li(\d+)ELi(\d+)ELi(\d+)E", name)
    cfg = "/".join(m.groups()) if m else "tiles:" + re.sub(r"^_Z\d*|I.*", "", name)[:24]
    key = (cfg, int(row["Grid_Size_X"]) // int(row["Workgroup_Size_X"]), int(row["Grid_Size_Y"]))
    d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    a = acc.setdefault(key, [])
    a.append(d)
for (cfg, gx, gy), ds in acc.items():
    ds = sorted(ds)[: max(1, len(ds) * 3 // 4)]            # drop the slowest quarter (first launches)
    print(f"{cfg:28s} grid {gx:4d} x {gy:3d}  n={len(ds):4d}  avg {sum(ds) / len(ds) / 1e3:7.2f} us  min {ds[0] / 1e3:6.2f}")
