import re,collections,sys,statistics
d=collections.defaultdict(list)
for l in open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/g7probe/probe.log'):
    m=re.match(r'(\S+) +ABL=(\d+)\s+(.*?)\s+M=(\d+) N=(\d+) K=(\d+) :\s+([\d.]+) us\s+([\d.]+) TFLOP',l)
    if m: d[(m.group(3),m.group(1)+'/A'+m.group(2))].append(float(m.group(7)))
names=[];cols=[]
for (n,a) in d:
    if n not in names: names.append(n)
    if a not in cols: cols.append(a)
print("min us     %-20s"%""+"".join("%10s"%c for c in cols))
for n in names: print("%-30s"%n+"".join(("%10.1f"%(min(d[(n,a)])) if d.get((n,a)) else "%10s"%"-") for a in cols))
print("median")
for n in names: print("%-30s"%n+"".join(("%10.1f"%(statistics.median(d[(n,a)])) if d.get((n,a)) else "%10s"%"-") for a in cols))
