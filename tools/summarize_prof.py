"""Condense the rocprofv3 output of tools/prof_round.sh (gpurun_out/prof_*) into the files under profiles/:
   profiles/<tag>_kernel_stats.csv  - per-kernel calls / total / average from --kernel-trace --stats
   profiles/r01_hbm_traffic.json    - HBM bytes per launch of the dominant GEMM kernel from the --pmc passes
Usage: python tools/summarize_prof.py <tag>        (e.g. r01_bench_v5)
"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")


def find(pattern):
    hits = glob.glob(os.path.join(OUT, pattern), recursive=True)      # gpurun merges into gpurun_out/: older rounds' files stay
    return max(hits, key=os.path.getmtime) if hits else None


def kernel_stats(tag):
    path = find("prof_stats/**/*kernel_stats.csv")
    if not path:
        print("no kernel_stats.csv under gpurun_out/prof_stats")
        return None
    rows = list(csv.DictReader(open(path)))
    dst = os.path.join(REPO, "profiles", tag + "_kernel_stats.csv")
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct"])
        for r in rows:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
    print("wrote", dst)
    return rows


def pmc(counter):
    path = find("prof_%s/**/*counter_collection.csv" % counter)
    if not path:
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01_bench"
    rnd = sys.argv[2] if len(sys.argv) > 2 else tag.split("_")[0]        # round prefix of the traffic file (r03 ...)
    rows = kernel_stats(tag)
    fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
    gemm = [k for k in fetch if "gemm_nt_kernel" in k]
    if not gemm:
        print("no GEMM kernel in the PMC passes")
        return
    # the dominant GEMM generation = the one with the most launches
    name = max(gemm, key=lambda k: len(fetch[k]))
    names = [k for k in gemm if k.split("<")[0] == name.split("<")[0] and k[:60].split(",")[0] == name[:60].split(",")[0]]
    f = [v for k in names for v in fetch[k]]
    w = [v for k in names for v in write.get(k, [])]
    favg, wavg = sum(f) / len(f), (sum(w) / len(w) if w else 0.0)
    # bert-base at 1024 x 128 tokens: the four contractions of a layer, bf16 in and out
    M = 1024 * 128
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    # N = 768 launches add the residual; since round 3 the bf16 residual stream has two planes (read hi + lo, write hi + lo)
    two = any("3, true>" in k or ", 3, " in k for k in names)
    compulsory = sum(2 * (M * k + n * k + M * n) + ((2 * M * n) * (3 if two else 1) if n == 768 else 0) for n, k in shapes) / len(shapes)
    per = {}
    for k in names:
        if fetch[k]:
            per[k[:72]] = {"launches": len(fetch[k]), "fetch_kb": sum(fetch[k]) / len(fetch[k]),
                           "write_kb": (sum(write[k]) / len(write[k])) if write.get(k) else None}
    out = {
        "kernel": name[:60],
        "launches": len(f),
        "fetch_size_kb_avg": favg,
        "write_size_kb_avg": wavg,
        "hbm_bytes_per_launch": 2 * favg * 1024 + wavg * 1024,
        "compulsory_bytes_per_launch": compulsory,
        "two_plane_residual_stream": two,
        "per_variant": per,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (128-B requests tallied at 64 B); WRITE_SIZE as reported "
                "(uncalibrated); separate rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1 --no-search`; "
                "compulsory = operands + output (+ residual where present) once, averaged over the layer's four GEMMs",
    }
    dst = os.path.join(REPO, "profiles", rnd + "_hbm_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst, json.dumps(out)[:300])
    if rows:
        for r in rows[:8]:
            print("%-70s calls=%s avg=%.1f us  %s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))


if __name__ == "__main__":
    main()
