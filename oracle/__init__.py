"""CPU oracle for the MI355X hot path — TEST INFRASTRUCTURE ONLY.

Nothing under `openmatch_amd/` or `openmatch/` may import this package: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and only as the
checker.  Each module restates, in plain torch/numpy CPU arithmetic, what the reference
(thunlp/OpenMatch v2, `/root/reference/src/openmatch`) and the third-party libraries it calls
(HF transformers 5.15 `BertModel`/`T5EncoderModel`, `faiss.IndexFlatIP`) compute on the
bi-encoder encode -> score/top-k -> contrastive-train path, citing file:line.

Pinning: the reference has no tests and no usable golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against the REFERENCE ITSELF executed in the
build container — `oracle/make_golden.py` imports `/root/reference/src/openmatch` and HF,
runs them on seeded inputs, checks every oracle function against their outputs and writes
the fixtures under `tests/golden/`.  `faiss` is not installable here (no network), so the
`IndexFlatIP` restatement is pinned only against its published semantics and an fp64
brute-force; DESIGN.md says "parity unpinned" for that one component.
"""
