"""CPU restatement of `faiss.IndexFlatIP` as the reference uses it
(src/openmatch/retriever/dense_retriever.py:38-41 ctor, :105 add, :135 reset, :180 search).

faiss is a C++ dependency that is neither vendored in /root/reference nor installable here
(README asks for faiss-cpu/faiss-gpu, unversioned; v1/requirements.txt:3 pins
faiss-cpu==1.6.3).  Published semantics restated: rows stored as float32 in insertion order;
`search(x, k)` returns D[q,j] = <x_q, y_I[q,j]> in float32, each row sorted by descending
score, I = int64 insertion indices, padded with I = -1 / D = -3.4028235e38 when ntotal < k.
Tie order is unspecified by faiss; this oracle breaks ties by ascending index.
PARITY UNPINNED against the real library (no golden vectors exist in the reference).
Test infrastructure only — see oracle/__init__.py.
"""
import numpy as np
import torch


class IndexFlatIP:
    def __init__(self, d):
        self.d = int(d)
        self._rows = []
        self.ntotal = 0

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d
        self._rows.append(x)
        self.ntotal += x.shape[0]

    def reset(self):
        self._rows = []
        self.ntotal = 0

    def _matrix(self):
        if len(self._rows) > 1:
            self._rows = [np.concatenate(self._rows)]
        return self._rows[0] if self._rows else np.zeros((0, self.d), np.float32)

    def search(self, x, k, dtype=torch.float32, block=262144):
        """Blocked matmul + top-k.  dtype=float64 gives the tie-adjudication ground truth."""
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dtype)
        Y = self._matrix()
        nq, n = x.shape[0], Y.shape[0]
        D = torch.full((nq, k), float(np.finfo(np.float32).min), dtype=dtype)
        I = torch.full((nq, k), -1, dtype=torch.int64)
        best_s = torch.empty((nq, 0), dtype=dtype)
        best_i = torch.empty((nq, 0), dtype=torch.int64)
        for s in range(0, n, block):
            yb = torch.from_numpy(Y[s:s + block]).to(dtype)
            sc = x @ yb.t()
            ids = torch.arange(s, s + yb.shape[0]).expand(nq, -1)
            best_s = torch.cat([best_s, sc], 1)
            best_i = torch.cat([best_i, ids], 1)
            if best_s.shape[1] > k:
                # stable descending sort => ties keep ascending index
                order = torch.sort(best_s, dim=1, descending=True, stable=True).indices[:, :k]
                best_s = torch.gather(best_s, 1, order)
                best_i = torch.gather(best_i, 1, order)
        if n > 0:
            order = torch.sort(best_s, dim=1, descending=True, stable=True).indices[:, :k]
            m = order.shape[1]
            D[:, :m] = torch.gather(best_s, 1, order)
            I[:, :m] = torch.gather(best_i, 1, order)
        return D.to(torch.float32).numpy() if dtype == torch.float32 else D.numpy(), I.numpy()


def topk_sets_equal(I_test, I_ref, D64_full_fn=None, rel_tol=1e-6):
    """Compare top-k id SETS per query; returns (n_exact, n_tie_only, n_bad, detail).
    A mismatch counts as 'tie only' when every disputed id's fp64 score is within rel_tol of the
    fp64 k-th score (two fp32 implementations legitimately disagree there: SURVEY.md 7.1)."""
    n_exact = n_tie = n_bad = 0
    detail = []
    for q in range(I_ref.shape[0]):
        a, b = set(I_test[q].tolist()), set(I_ref[q].tolist())
        if a == b:
            n_exact += 1
            continue
        disputed = sorted((a ^ b) - {-1})
        ok = False
        if D64_full_fn is not None and disputed:
            sc, kth = D64_full_fn(q, disputed)
            ok = bool(np.all(np.abs(sc - kth) <= rel_tol * (1.0 + abs(kth))))
        if ok:
            n_tie += 1
        else:
            n_bad += 1
            detail.append((q, disputed[:8]))
    return n_exact, n_tie, n_bad, detail
