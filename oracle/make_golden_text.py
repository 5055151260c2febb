"""Generates tests/golden/text/* by EXECUTING THE REFERENCE's text feed in the build container:
`openmatch.dataset.InferenceDataset` (tsv, every rank of a 2-process partition, final and
non-final tokenisation) and `openmatch.dataset.DRTrainDataset / RRTrainDataset` (epochs 0..3, with
and without the seeded selection).  Run:  cd /tmp && python /root/repo/oracle/make_golden_text.py

The tokenizer is a `BertTokenizer` over a small word-piece vocabulary written here (no pretrained
tokenizer exists offline).  The fixture keeps the input files and the reference's outputs, so the
tests need neither /root/reference nor `datasets`.
"""
import json
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, os.path.join(REF, "src"))

import numpy as np  # noqa: E402
import datasets  # noqa: E402,F401
from transformers import BertTokenizer  # noqa: E402

faiss_stub = types.ModuleType("faiss")            # the reference package imports faiss at import time
faiss_stub.IndexFlatIP = object
sys.modules["faiss"] = faiss_stub
import openmatch  # noqa: E402
assert openmatch.__file__.startswith(REF), openmatch.__file__
from openmatch.dataset import DRTrainDataset, InferenceDataset, RRTrainDataset  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "text")
NS = types.SimpleNamespace
WORDS = ("the of and a in to is for on with as by an be this that from at are it was or which retrieval dense "
         "passage query model index search vector encoder training negative positive score rank gpu kernel "
         "memory tile wave matrix sparse neural language open match marco document title text question answer "
         "##s ##ing ##ed ##er ##ly ##tion").split()


def v4_encode_plus(tok):
    """transformers >= 5 dropped `encode_plus` / `prepare_for_model` for lists of ids, which the
    reference's training feed calls (train_dataset.py:60-67,139-147; `transformers>=4.10` unpinned in
    setup.py).  What 4.x did for a single id list with padding off: special tokens around the ids,
    cut to max_length.  The shim restores exactly that so the reference's example-building logic runs."""
    def encode_plus(ids, truncation=None, max_length=None, padding=False, return_attention_mask=False,
                    return_token_type_ids=False):
        room = max_length - 2
        out = {"input_ids": [tok.cls_token_id] + list(ids)[:room] + [tok.sep_token_id]}
        if padding == "max_length":      # the re-ranker's call (reranker.py:23-29): 4.x pads and returns all three
            n = len(out["input_ids"])
            out = {"input_ids": out["input_ids"] + [tok.pad_token_id] * (max_length - n),
                   "token_type_ids": [0] * max_length, "attention_mask": [1] * n + [0] * (max_length - n)}
        return out
    return encode_plus


def write_inputs():
    os.makedirs(OUT, exist_ok=True)
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS + list("abcdefghijklmnopqrstuvwxyz0123456789") \
        + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"] + [",", ".", "?", "-"]
    vocab = list(dict.fromkeys(vocab))          # "a", "##s" occur twice above; ids must be dense
    open(os.path.join(OUT, "vocab.txt"), "w").write("\n".join(vocab) + "\n")
    rng = np.random.default_rng(20260925)
    plain = [w for w in WORDS if not w.startswith("##")]
    sent = lambda n: " ".join(rng.choice(plain, size=n))
    with open(os.path.join(OUT, "corpus.tsv"), "w") as f:          # id, title, text
        for i in range(23):
            f.write("d%d\t%s\t%s\n" % (100 + i, sent(int(rng.integers(1, 4))), sent(int(rng.integers(3, 30)))))
    # queries as tsv too: the reference's JsonlDataset cannot be constructed (inference_dataset.py:129
    # calls the base __init__ without tokenizer / data_args -> TypeError), so json input is unpinnable
    qs = [("q%d" % i, sent(int(rng.integers(2, 12)))) for i in range(9)]
    with open(os.path.join(OUT, "queries.tsv"), "w") as f:
        for qid, text in qs:
            f.write("%s\t%s\n" % (qid, text))
    with open(os.path.join(OUT, "queries.json"), "w") as f:        # same records, one object per line
        for qid, text in qs:
            f.write(json.dumps({"id": qid, "text": text}) + "\n")
    tok = BertTokenizer(os.path.join(OUT, "vocab.txt"))
    if not hasattr(tok, "prepare_for_model"):
        object.__setattr__(tok, "encode_plus", v4_encode_plus(tok))
    ids = lambda n: tok.encode(sent(n), add_special_tokens=False)
    with open(os.path.join(OUT, "train.jsonl"), "w") as f:
        for i in range(7):
            f.write(json.dumps({"query": ids(int(rng.integers(2, 8))),
                                "positives": [ids(int(rng.integers(4, 40))) for _ in range(int(rng.integers(1, 4)))],
                                "negatives": [ids(int(rng.integers(4, 40))) for _ in range(int(rng.integers(2, 9)))]}) + "\n")
    return tok


def data_args(**kw):
    base = dict(corpus_path=os.path.join(OUT, "corpus.tsv"), query_path=os.path.join(OUT, "queries.tsv"),
                processed_data_path=None, q_max_len=8, p_max_len=24, dataset_proc_num=1,
                query_template="<text>", doc_template="<title> [SEP] <text>",
                query_column_names="id,text", doc_column_names="id,title,text",
                train_path=os.path.join(OUT, "train.jsonl"), train_dir=None, eval_path=None, train_n_passages=4,
                positive_passage_no_shuffle=False, negative_passage_no_shuffle=False)
    base.update(kw)
    return NS(**base)


def plain(rec):
    return {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in dict(rec).items()}


def main():
    tok = write_inputs()
    golden = {"inference": [], "train": []}
    for is_query in (False, True):
        for final in (True, False):
            for nproc, bs in ((1, 4), (2, 3)):
                for rank in range(nproc):
                    ds = InferenceDataset.load(tok, data_args(), is_query=is_query, final=final, stream=True,
                                               batch_size=bs, num_processes=nproc, process_index=rank)
                    golden["inference"].append({"is_query": is_query, "final": final, "num_processes": nproc,
                                                "batch_size": bs, "process_index": rank,
                                                "records": [plain(r) for r in ds]})
    # non-stream random access (the re-ranker's use)
    ds = InferenceDataset.load(tok, data_args(), is_query=False, final=False, stream=False)
    golden["getitem"] = {k: plain(ds[k]) for k in ("d100", "d111", "d122")}
    for cls, name in ((DRTrainDataset, "dr"), (RRTrainDataset, "rr")):
        for seed in (None, 13):
            for flags in ({}, {"positive_passage_no_shuffle": True, "negative_passage_no_shuffle": True}):
                for epoch in range(4):
                    if seed is None and epoch:
                        continue
                    # shuffle_seed=None keeps file order; the seeded CHOICES come from trainer.args.seed
                    trainer = None if seed is None else NS(state=NS(epoch=float(epoch)), args=NS(seed=seed))
                    ds = cls(tok, data_args(**flags), trainer=trainer, shuffle_seed=None)
                    if seed is None:      # the reference reads trainer.state unconditionally in __iter__
                        it = iter(ds.dataset.map(ds.get_process_fn(0, None), remove_columns=["positives", "negatives"]))
                    else:
                        it = iter(ds)
                    rows = [json.loads(json.dumps(dict(ex), default=lambda o: dict(o))) for ex in it]
                    golden["train"].append({"kind": name, "seed": seed, "flags": flags, "epoch": epoch, "examples": rows})
    # the re-ranker's pair encoding: the reference's own encode_pair (retriever/reranker.py:23-29) on id lists
    # short enough to pad and long enough to truncate
    from openmatch.retriever.reranker import encode_pair
    rng = np.random.default_rng(7)
    golden["encode_pair"] = []
    for nq, nd, m1, m2 in ((3, 5, 8, 24), (8, 24, 8, 24), (12, 40, 8, 24), (1, 1, 32, 128), (30, 200, 32, 128)):
        q = [int(x) for x in rng.integers(5, 60, size=nq)]
        d = [int(x) for x in rng.integers(5, 60, size=nd)]
        golden["encode_pair"].append({"q": q, "d": d, "max_len_1": m1, "max_len_2": m2,
                                      "out": plain(encode_pair(tok, q, d, m1, m2))})
    json.dump(golden, open(os.path.join(OUT, "reference_outputs.json"), "w"))
    print("wrote", OUT, {k: len(v) for k, v in golden.items()})


if __name__ == "__main__":
    main()
