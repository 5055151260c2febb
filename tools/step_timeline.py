#!/usr/bin/env python
"""Timeline of ONE step out of a rocprofv3 --kernel-trace csv: start (us), duration, queue, short kernel name, grid -- from the
second-to-last launch of MARK (default embed_kernel: the first kernel of a training forward) to the last one.
  python tools/step_timeline.py gpurun_out/<tag>/stats/*/*_kernel_trace.csv [MARK]"""
import csv
import sys

SHORT = ("gemm_nt_kernel2", "gemm_tn_wide", "kernel7r16", "kernel7c16", "gemm_nt_kernel7<", "gemm_nt_kernel6", "ln_bwd", "attention_bwd16",
         "attention_fwd16", "layernorm_bf16x8", "adamw", "grad_sqnorm", "sqnorm_reduce", "transpose_batch", "embed_kernel", "dropout_kernel",
         "FillFunctor", "pool_bwd", "pool_kernel", "ce_rows", "ce_reduce", "dq_kernel", "dp_kernel", "gemm_nt_kernel<float", "copyBuffer",
         "splitk", "reduce_ln", "multi_tensor_apply")


def short(n):
    for k in SHORT:
        if k in n:
            return k
    return n[:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    mark = sys.argv[2] if len(sys.argv) > 2 else "embed_kernel"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
    s, e = idx[-2], idx[-1]
    t0 = int(rows[s]["Start_Timestamp"])
    print("step span %.1f us, %d kernels" % ((int(rows[e]["Start_Timestamp"]) - t0) / 1e3, e - s))
    busy, prev_end = 0.0, 0.0
    for r in rows[s:e]:
        st, en = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        gap = st - prev_end
        print("%9.1f %7.1f  %s q%s %s %s" % (st, en - st, ("gap %6.1f" % gap) if gap > 8 else " " * 10, r["Queue_Id"], short(r["Kernel_Name"]), r["Grid_Size_X"]))
        busy += max(0.0, en - max(st, prev_end))
        prev_end = max(prev_end, en)
    print("device busy %.1f us of the span" % busy)


if __name__ == "__main__":
    main()
