"""TEST INFRASTRUCTURE ONLY (imported by tests/ and bench.py's parity legs, never by openmatch_amd/).

Independent check of `IndexFlatIP.search` results at sizes the CPU oracle (oracle/flatip.py) cannot finish in
seconds -- the benchmark's own 8 841 823 x 768, k = 1000 (reference call site: retriever/dense_retriever.py:180).
Semantics restated from oracle/flatip.py: D[q, j] = <x_q, y_I[q, j]> in fp32, top-k by descending score; two fp32
implementations may legitimately disagree on ids whose fp64 scores tie with the k-th score, so a
differing id set is adjudicated in fp64: every disputed id must lie within rel_tol of the fp64 k-th score.

The checker is plain torch on the device (chunked fp32 `matmul` + `topk`, fp64 `matmul` for adjudication): none of the
product's kernels are involved."""
import torch


def reference_topk(rows: torch.Tensor, queries: torch.Tensor, k: int, chunk: int = 1 << 20):
    """Chunked exact fp32 search: rows [n, d] f32 (device), queries [q, d] f32 -> (D [q, k], I [q, k]) sorted descending."""
    n = rows.shape[0]
    best_v = best_i = None
    for s in range(0, n, chunk):
        sc = queries @ rows[s:s + chunk].t()
        v, i = torch.topk(sc, min(k, sc.shape[1]), dim=1)
        i = i + s
        if best_v is None:
            best_v, best_i = v, i
        else:
            cv, ci = torch.cat([best_v, v], 1), torch.cat([best_i, i], 1)
            best_v, sel = torch.topk(cv, min(k, cv.shape[1]), dim=1)
            best_i = torch.gather(ci, 1, sel)
        del sc
    return best_v, best_i


def kth_score_fp64(rows: torch.Tensor, query: torch.Tensor, k: int, chunk: int = 1 << 20):
    """The exact fp64 k-th largest inner product of one query against all rows (chunked)."""
    q64 = query.double()
    best = None
    for s in range(0, rows.shape[0], chunk):
        sc = rows[s:s + chunk].double() @ q64
        v = torch.topk(sc, min(k, sc.numel())).values
        best = v if best is None else torch.topk(torch.cat([best, v]), min(k, best.numel() + v.numel())).values
    return best[-1].item()


def adjudicate(rows: torch.Tensor, queries: torch.Tensor, I_test: torch.Tensor, I_ref: torch.Tensor, k: int, rel_tol: float = 2e-6):
    """Per query: id sets identical, or every disputed id within rel_tol of the fp64 k-th score ('tie only'), or wrong.
    Returns (n_exact, n_tie_only, n_bad, detail)."""
    n_exact = n_tie = n_bad = 0
    detail = []
    It, Ir = I_test.cpu(), I_ref.cpu()
    for qi in range(Ir.shape[0]):
        a, b = set(It[qi].tolist()), set(Ir[qi].tolist())
        if a == b:
            n_exact += 1
            continue
        disputed = sorted((a ^ b) - {-1})
        kth = kth_score_fp64(rows, queries[qi], k)
        idx = torch.tensor(disputed, device=rows.device)
        sc = (rows[idx].double() @ queries[qi].double()).cpu()
        if bool(((sc - kth).abs() <= rel_tol * (1.0 + abs(kth))).all()):
            n_tie += 1
        else:
            n_bad += 1
            detail.append((qi, disputed[:8]))
    return n_exact, n_tie, n_bad, detail
