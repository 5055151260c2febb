"""Fixtures at the shapes that are actually BENCHMARKED (full-size models), produced by EXECUTING THE
REFERENCE in the build container:  python oracle/make_golden_base.py   (needs /root/reference).

  config1_bert_base.npz   BASELINE config 1 / SURVEY 8(d) row 1: seeded BertConfig() (12 x 768), 1 000 passages
                          x 128 tok + 100 queries x 32 tok  ->  reference DRModelForInference  ->  reference
                          Retriever.search (oracle.flatip injected as `faiss`)  ->  save_as_trec  ->  eval_mrr,
                          once in fp32 and once under torch.autocast("cpu", bfloat16) -- the reference's own
                          16-bit mode (its `--fp16` switch wraps the model call in autocast,
                          retriever/dense_retriever.py:76) -- so that the HIP bf16 path is judged against what
                          the REFERENCE loses in 16-bit arithmetic, not against an arbitrary threshold.
  config1_spread.npz      The same chain with BertConfig(initializer_range=0.1) (round 4): with the default 0.02 every CLS
                          vector of a random-init bert-base is nearly the same (all dots 762 +- 0.3), so ONE swapped
                          near-tie moves MRR@10 by 0.005 and the "MRR@10 within 1e-4" gate cannot be evaluated at any
                          16-bit precision, the reference's own included (0.0035).  Five-fold weights spread the scores
                          (std of a query's 1 000 dots ~ 1.7e-2 of the dot scale instead of 4e-4) -- and raise the 16-bit
                          noise with them: the reference's own float16 autocast run (stored: P_ac16, Q_ac16) is ~1e-3 of the
                          dot scale from its fp32 run.  The relevant document of a query is drawn from the reference ranks
                          1..10 whose document is separated from both neighbours by more than 2.5 x THAT noise; a query
                          without one gets its judgment at rank 50 (0 in MRR@10 for every run).  The gate then tests rank
                          stability at the reference's own 16-bit noise level, not tie-breaking.
  gtr_base.npz            GTR-base-sized T5 encoder (12 x 768, relu, mean pooling, 768->768 head, normalised):
                          64 passages + 16 queries, fp32 and autocast.
  bert_large_rr.npz       bert-large cross-encoder (24 x 1024) RRModel scores of 32 pairs x 162 tok, fp32 and autocast.

  train_base.npz          The BENCHMARKED training step (round 5): bert-base width (768 / 3072 / 12 heads), two layers, 8 queries
                          x 32 tok + 64 passages x 128 tok (train_n_passages = 8, the per-GPU batch of docs/dr-msmarco-passage.md:
                          75-76), dropout 0  ->  reference DRModel.forward + loss.backward() three times: fp32, under
                          torch.autocast(bfloat16) and under torch.autocast(float16) with a static loss scale (what the reference's
                          `--fp16` training does through HF Trainer's GradScaler, trainer/dense_trainer.py:141-149).  Stored: the fp32
                          gradients on a strided row subset of every parameter (stride 11: every residue of a 16 / 32 / 256-row tile
                          is visited; the embedding table on the rows of tokens that occur), and PER TENSOR the relative L2 distance
                          of the two autocast runs from the fp32 run on that subset and on the whole tensor -- the yardstick the
                          HIP 16-bit training step is held to (factor 1.0).

Weights are re-created from the seed by the tests (HF is in the image); a checksum pins them.
"""
import os
import pickle
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as mg  # noqa: E402  (puts /root/reference/src first on sys.path, injects the faiss stub)
from oracle import flatip  # noqa: E402

from transformers import BertConfig, BertModel, T5Config, T5EncoderModel  # noqa: E402
from openmatch.modeling import DRModelForInference  # noqa: E402  (the reference)
from openmatch.modeling.linear import LinearHead  # noqa: E402
from openmatch.modeling.reranking_model import RRModel  # noqa: E402
from openmatch.retriever import Retriever  # noqa: E402
from openmatch.utils import save_as_trec  # noqa: E402

NS = mg.NS
OUT = mg.OUT
SEED = 20260925


def checksum(model):
    sd = model.state_dict()
    return np.array([float(sum(v.double().sum() for v in sd.values())), float(sum(v.double().abs().sum() for v in sd.values()))])


def lengths(mask):
    return mask.sum(1).astype(np.int16)


def encode_all(ref, which, ids, mask, bs, autocast, extra=None, ac_dtype=torch.bfloat16):
    """The reference's batch loop (dense_retriever.py:70-83): model(passage=batch) / model(query=batch), optional
    autocast, .cpu().numpy() per batch, concatenate."""
    outs = []
    t0 = time.time()
    for s in range(0, ids.shape[0], bs):
        batch = {"input_ids": torch.from_numpy(ids[s:s + bs]), "attention_mask": torch.from_numpy(mask[s:s + bs])}
        if extra is not None:
            batch.update({k: torch.from_numpy(v[s:s + bs]) for k, v in extra.items()})
        with torch.no_grad():
            if autocast:
                with torch.autocast("cpu", dtype=ac_dtype):
                    o = ref(**{which: batch})
            else:
                o = ref(**{which: batch})
        reps = o.p_reps if which == "passage" else o.q_reps
        outs.append(reps.float().cpu().numpy())
    print(f"    {which} x{ids.shape[0]} autocast={autocast}: {time.time() - t0:.1f} s", flush=True)
    return np.concatenate(outs)


def reference_search(P, Q, doc_ids, qry_ids, k):
    """Reference Retriever.search over shard pickles (two corpus shards, two query shards)."""
    n, nq = P.shape[0], Q.shape[0]
    with tempfile.TemporaryDirectory() as tmp:
        for r in range(2):
            with open(os.path.join(tmp, f"embeddings.query.rank.{r}"), "wb") as f:
                pickle.dump((Q[r * nq // 2:(r + 1) * nq // 2], qry_ids[r * nq // 2:(r + 1) * nq // 2]), f, protocol=4)
        args = NS(device="cpu", output_dir=tmp, world_size=2, process_index=0, local_process_index=0)
        r = Retriever(torch.nn.Linear(1, 1), None, args)
        r.index = flatip.IndexFlatIP(P.shape[1])
        r.doc_lookup = []
        for rk in range(2):
            r.index.add(P[rk * n // 2:(rk + 1) * n // 2])
            r.doc_lookup.extend(doc_ids[rk * n // 2:(rk + 1) * n // 2])
        run = r.search(k)
        trec = os.path.join(tmp, "run.trec")
        save_as_trec(run, trec)
        text = open(trec).read()
    return run, text


def run_to_arrays(run, qry_ids, doc_ids):
    pos = {d: i for i, d in enumerate(doc_ids)}
    I = np.array([[pos[d] for d in run[q]] for q in qry_ids], np.int32)
    D = np.array([[run[q][d] for d in run[q]] for q in qry_ids], np.float32)
    return I, D


def config1(rng):
    torch.manual_seed(0)
    cfg = BertConfig()
    lm = BertModel(cfg).eval()
    ref = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False))
    n, nq = 1000, 100
    p_ids, p_mask = mg.synth_batch(rng, n, 128, cfg.vocab_size, 16)
    q_ids, q_mask = mg.synth_batch(rng, nq, 32, cfg.vocab_size, 4)
    doc_ids = [f"D{7 * i + 3}" for i in range(n)]
    qry_ids = [f"Q{i}" for i in range(nq)]
    out = {"p_input_ids": p_ids.astype(np.uint16), "p_len": lengths(p_mask), "q_input_ids": q_ids.astype(np.uint16),
           "q_len": lengths(q_mask), "doc_ids": np.array(doc_ids), "qry_ids": np.array(qry_ids), "weight_checksum": checksum(lm)}
    eval_mrr = mg.reference_eval_mrr()
    runs = {}
    for tag, ac in (("f32", False), ("ac", True)):
        P = encode_all(ref, "passage", p_ids, p_mask, 50, ac)
        Q = encode_all(ref, "query", q_ids, q_mask, 50, ac)
        run, trec = reference_search(P, Q, doc_ids, qry_ids, 100)
        runs[tag] = run
        I, D = run_to_arrays(run, qry_ids, doc_ids)
        out[f"P_{tag}"] = P if tag == "f32" else P.astype(np.float16)
        out[f"Q_{tag}"] = Q if tag == "f32" else Q.astype(np.float16)
        out[f"I100_{tag}"], out[f"D100_{tag}"] = I, D
        if tag == "f32":
            out["trec_f32"] = np.array(trec)
            # qrels: one relevant document per query at reference rank r ~ U{1..20}
            qrel = {}
            for qid in qry_ids:
                ranked = sorted(run[qid].items(), key=lambda kv: kv[1], reverse=True)
                qrel[qid] = {ranked[int(rng.integers(0, 20))][0]: 1}
            out["qrel_docs"] = np.array([list(qrel[q])[0] for q in qry_ids])
        out[f"mrr10_{tag}"] = np.array(eval_mrr(qrel, run, cutoff=10)["all"])
    # how far the reference's own 16-bit mode is from its fp32 mode (the yardstick for the HIP bf16 path)
    Pf, Pa = torch.from_numpy(out["P_f32"]).double(), torch.from_numpy(out["P_ac"].astype(np.float32)).double()
    Qf, Qa = torch.from_numpy(out["Q_f32"]).double(), torch.from_numpy(out["Q_ac"].astype(np.float32)).double()
    cos = torch.nn.functional.cosine_similarity(Pf, Pa, dim=1)
    dd = ((Qa @ Pa.t()) - (Qf @ Pf.t())).abs()
    ov = [len(set(a.tolist()) & set(b.tolist())) for a, b in zip(out["I100_f32"], out["I100_ac"])]
    out["ac_vs_f32"] = np.array([float(cos.min()), float(cos.mean()), float(dd.max()), float(np.mean(ov)), float(np.min(ov))])
    print("  reference autocast-bf16 vs reference fp32: min cos %.6f  mean cos %.6f  max|ddot| %.4f  top-100 overlap mean %.1f min %d"
          % tuple(out["ac_vs_f32"]), " MRR@10 f32 %.4f  autocast %.4f" % (float(out["mrr10_f32"]), float(out["mrr10_ac"])))
    np.savez_compressed(os.path.join(OUT, "config1_bert_base.npz"), **out)
    print("wrote config1_bert_base.npz")


def config1_ac16():
    """Adds the reference's float16-autocast run (its real `--fp16` mode, retriever/dense_retriever.py:76) to an existing
    config1_bert_base.npz: embeddings rounded to f16 (P_ac16, Q_ac16), its top-100 ranking and its MRR@10 on the fixture's judgments
    -- the yardstick a float16 path's MRR deviation belongs next to (round 5).  The stored fp32 / bf16-autocast arrays stay as they
    are (bf16 autocast on the CPU is not bit-reproducible across runs; the fp32 embeddings are, and are re-checked here)."""
    path = os.path.join(OUT, "config1_bert_base.npz")
    g = np.load(path, allow_pickle=False)
    out = {k: g[k] for k in g.files}
    torch.manual_seed(0)
    cfg = BertConfig()
    lm = BertModel(cfg).eval()
    assert np.allclose(checksum(lm), out["weight_checksum"], rtol=1e-12)
    ref = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False))
    L = lambda ids, lens, n: (ids.astype(np.int64), (np.arange(n)[None, :] < lens.astype(np.int64)[:, None]).astype(np.int64))
    p_ids, p_mask = L(out["p_input_ids"], out["p_len"], 128)
    q_ids, q_mask = L(out["q_input_ids"], out["q_len"], 32)
    doc_ids, qry_ids = [str(x) for x in out["doc_ids"]], [str(x) for x in out["qry_ids"]]
    Qf = encode_all(ref, "query", q_ids, q_mask, 50, False)
    assert np.array_equal(Qf, out["Q_f32"]), "the fp32 query embeddings of this run differ from the stored ones"
    P16 = encode_all(ref, "passage", p_ids, p_mask, 50, True, ac_dtype=torch.float16)
    Q16 = encode_all(ref, "query", q_ids, q_mask, 50, True, ac_dtype=torch.float16)
    run16, _ = reference_search(P16, Q16, doc_ids, qry_ids, 100)
    I16, D16 = run_to_arrays(run16, qry_ids, doc_ids)
    qrel = {q: {str(d): 1} for q, d in zip(qry_ids, out["qrel_docs"])}
    out["P_ac16"], out["Q_ac16"], out["I100_ac16"] = P16.astype(np.float16), Q16.astype(np.float16), I16
    out["mrr10_ac16"] = np.array(mg.reference_eval_mrr()(qrel, run16, cutoff=10)["all"])
    Pf, Pa = torch.from_numpy(out["P_f32"]).double(), torch.from_numpy(P16).double()
    Qd, Qa = torch.from_numpy(out["Q_f32"]).double(), torch.from_numpy(Q16).double()
    cos = torch.nn.functional.cosine_similarity(Pf, Pa, dim=1)
    dd = ((Qa @ Pa.t()) - (Qd @ Pf.t())).abs()
    ov = [len(set(a.tolist()) & set(b.tolist())) for a, b in zip(out["I100_f32"], I16)]
    out["ac16_vs_f32"] = np.array([float(cos.min()), float(cos.mean()), float(dd.max()), float(np.mean(ov)), float(np.min(ov))])
    print("  config 1, reference float16 autocast vs its fp32: min cos %.8f mean cos %.8f max|ddot| %.4f top-100 overlap %.1f (min %d); MRR@10 fp32 %.4f, "
          "float16 autocast %.4f (|d| %.4f; bf16 autocast: %.4f)" % (*out["ac16_vs_f32"], float(out["mrr10_f32"]), float(out["mrr10_ac16"]),
                                                                      abs(float(out["mrr10_ac16"]) - float(out["mrr10_f32"])), abs(float(out["mrr10_ac"]) - float(out["mrr10_f32"]))))
    np.savez_compressed(path, **out)
    print("updated config1_bert_base.npz")


def config1_spread(rng):
    torch.manual_seed(0)
    cfg = BertConfig(initializer_range=0.1)
    lm = BertModel(cfg).eval()
    ref = DRModelForInference(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False))
    n, nq = 1000, 100
    p_ids, p_mask = mg.synth_batch(rng, n, 128, cfg.vocab_size, 16)
    q_ids, q_mask = mg.synth_batch(rng, nq, 32, cfg.vocab_size, 4)
    doc_ids = [f"D{7 * i + 3}" for i in range(n)]
    qry_ids = [f"Q{i}" for i in range(nq)]
    out = {"p_input_ids": p_ids.astype(np.uint16), "p_len": lengths(p_mask), "q_input_ids": q_ids.astype(np.uint16),
           "q_len": lengths(q_mask), "doc_ids": np.array(doc_ids), "qry_ids": np.array(qry_ids), "weight_checksum": checksum(lm)}
    eval_mrr = mg.reference_eval_mrr()
    P = encode_all(ref, "passage", p_ids, p_mask, 50, False)
    Q = encode_all(ref, "query", q_ids, q_mask, 50, False)
    run, trec = reference_search(P, Q, doc_ids, qry_ids, 100)
    I, D = run_to_arrays(run, qry_ids, doc_ids)
    out.update(P_f32=P, Q_f32=Q, I100_f32=I, D100_f32=D)
    S = Q.astype(np.float64) @ P.astype(np.float64).T
    scale = float(np.abs(S).max())
    # the reference's OWN float16 mode on this model (torch.autocast float16 = its --fp16, retriever/dense_retriever.py:76):
    # the yardstick for the HIP float16 path, and the noise level the relevance judgments must clear
    P16 = encode_all(ref, "passage", p_ids, p_mask, 50, True, ac_dtype=torch.float16)
    Q16 = encode_all(ref, "query", q_ids, q_mask, 50, True, ac_dtype=torch.float16)
    run16, _ = reference_search(P16, Q16, doc_ids, qry_ids, 100)
    I16, _ = run_to_arrays(run16, qry_ids, doc_ids)
    S16 = Q16.astype(np.float64) @ P16.astype(np.float64).T
    noise = float(np.abs(S16 - S).max())
    cos16 = torch.nn.functional.cosine_similarity(torch.from_numpy(P).double(), torch.from_numpy(P16).double(), dim=1)
    ov16 = [len(set(a.tolist()) & set(b.tolist())) for a, b in zip(I, I16)]
    thr = 2.5 * noise
    qrel, gaps, in_top10 = {}, [], 0
    for qi, qid in enumerate(qry_ids):
        ranked = sorted(run[qid].items(), key=lambda kv: kv[1], reverse=True)
        sc = np.sort(S[qi])[::-1]
        gap_at = lambda r: min(sc[r - 1] - sc[r] if r else np.inf, sc[r] - sc[r + 1])
        ok = [r for r in range(10) if gap_at(r) > thr]
        if ok:
            r = int(ok[int(rng.integers(0, len(ok)))])
            in_top10 += 1
            gaps.append(gap_at(r) / scale)
        else:                       # no top-10 document clears the 16-bit noise: a judgment at rank 50 (0 in MRR@10 for every run)
            r = 49
        qrel[qid] = {ranked[r][0]: 1}
    out["qrel_docs"] = np.array([list(qrel[q])[0] for q in qry_ids])
    out["mrr10_f32"] = np.array(eval_mrr(qrel, run, cutoff=10)["all"])
    out["mrr10_ac16"] = np.array(eval_mrr(qrel, run16, cutoff=10)["all"])
    out["dot_scale"] = np.array(scale)
    out["qrel_min_gap_rel"] = np.array(min(gaps))
    out["qrel_in_top10"] = np.array(in_top10)
    out["score_std_rel"] = np.array(float(S.std(axis=1).mean() / scale))
    out["P_ac16"], out["Q_ac16"], out["I100_ac16"] = P16.astype(np.float16), Q16.astype(np.float16), I16
    # reference float16 autocast vs reference fp32: min cos, max |ddot|, top-100 overlap mean / min
    out["ac16_vs_f32"] = np.array([float(cos16.min()), noise, float(np.mean(ov16)), float(np.min(ov16))])
    print("  spread fixture: dot scale %.1f, per-query std of dots %.2e of it; reference float16 autocast vs its fp32: min cos %.8f, max|ddot| %.4f = %.2e of the scale, "
          "top-100 overlap %.1f (min %d); %d of %d queries have a top-10 relevant document separated by > %.2e of the scale; MRR@10 fp32 %.4f, float16 autocast %.4f"
          % (scale, float(out["score_std_rel"]), float(cos16.min()), noise, noise / scale, np.mean(ov16), np.min(ov16), in_top10, nq, thr / scale,
             float(out["mrr10_f32"]), float(out["mrr10_ac16"])))
    _spread_uniform_qrels(out, run, run16, qry_ids)
    np.savez_compressed(os.path.join(OUT, "config1_spread.npz"), **out)
    print("wrote config1_spread.npz")


def _spread_uniform_qrels(out, run, run16, qry_ids):
    """A SECOND set of judgments on the spread fixture, drawn WITHOUT looking at score gaps (round 5): one relevant
    document per query at a uniform reference rank in 1..10.  On these the reference's own float16 run does move MRR@10 (stored:
    mrr10_ac16_uniform) -- the unconditioned yardstick next to the gap-conditioned gate above, which passes by construction for
    anything at the reference's 16-bit noise level."""
    rng = np.random.default_rng(SEED + 16)
    eval_mrr = mg.reference_eval_mrr()
    qrel = {}
    for qid in qry_ids:
        ranked = sorted(run[qid].items(), key=lambda kv: kv[1], reverse=True)
        qrel[qid] = {ranked[int(rng.integers(0, 10))][0]: 1}
    out["qrel_docs_uniform"] = np.array([list(qrel[q])[0] for q in qry_ids])
    out["mrr10_f32_uniform"] = np.array(eval_mrr(qrel, run, cutoff=10)["all"])
    out["mrr10_ac16_uniform"] = np.array(eval_mrr(qrel, run16, cutoff=10)["all"])
    print("  spread fixture, unconditioned judgments (uniform rank 1..10): MRR@10 fp32 %.6f, reference float16 autocast %.6f (|d| %.6f)"
          % (float(out["mrr10_f32_uniform"]), float(out["mrr10_ac16_uniform"]), abs(float(out["mrr10_f32_uniform"]) - float(out["mrr10_ac16_uniform"]))))


def spread_qrels():
    """Adds the unconditioned judgments to an existing config1_spread.npz from the reference embeddings stored in it (the
    reference's Retriever.search is re-run on them: the same runs config1_spread() builds, without the two encoding passes)."""
    path = os.path.join(OUT, "config1_spread.npz")
    g = np.load(path, allow_pickle=False)
    out = {k: g[k] for k in g.files}
    doc_ids, qry_ids = [str(x) for x in out["doc_ids"]], [str(x) for x in out["qry_ids"]]
    run, _ = reference_search(out["P_f32"], out["Q_f32"], doc_ids, qry_ids, 100)
    I, _ = run_to_arrays(run, qry_ids, doc_ids)
    assert np.array_equal(I, out["I100_f32"]), "the stored fp32 embeddings do not reproduce the stored run"
    # the float16-autocast embeddings were stored rounded to 16 bits, so its run is rebuilt from the stored ranking itself
    # (eval_mrr only orders a query's documents by score)
    run16 = {q: {doc_ids[d]: -float(r) for r, d in enumerate(row)} for q, row in zip(qry_ids, out["I100_ac16"])}
    assert abs(mg.reference_eval_mrr()({q: {str(d): 1} for q, d in zip(qry_ids, out["qrel_docs"])}, run16, cutoff=10)["all"]
               - float(out["mrr10_ac16"])) < 1e-12
    _spread_uniform_qrels(out, run, run16, qry_ids)
    np.savez_compressed(path, **out)
    print("updated config1_spread.npz")


def gtr_base(rng):
    torch.manual_seed(1)
    cfg = T5Config(d_model=768, d_ff=3072, num_layers=12, num_heads=12, d_kv=64, feed_forward_proj="relu")
    lm = T5EncoderModel(cfg).eval()
    head = LinearHead(768, 768)
    ref = DRModelForInference(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True,
                              model_args=NS(encoder_only=True))
    p_ids, p_mask = mg.synth_batch(rng, 64, 128, cfg.vocab_size, 16, bert=False)
    q_ids, q_mask = mg.synth_batch(rng, 16, 32, cfg.vocab_size, 4, bert=False)
    out = {"p_input_ids": p_ids.astype(np.uint16), "p_len": lengths(p_mask), "q_input_ids": q_ids.astype(np.uint16),
           "q_len": lengths(q_mask), "weight_checksum": checksum(lm),
           "head_checksum": np.array([float(head.linear.weight.double().sum()), float(head.linear.weight.double().abs().sum())])}
    for tag, ac in (("f32", False), ("ac", True)):
        out[f"P_{tag}"] = encode_all(ref, "passage", p_ids, p_mask, 32, ac)
        out[f"Q_{tag}"] = encode_all(ref, "query", q_ids, q_mask, 16, ac)
    # the reference's REAL 16-bit mode is float16 autocast (retriever/dense_retriever.py:76,151: torch.cuda.amp.autocast());
    # round 5 records it next to the bf16-autocast run.  T5 in float16 can overflow (HF clamps, modeling_t5.py:
    # 467-474): non-finite rows are counted and the yardstick is taken over the finite ones.
    out["P_ac16"] = encode_all(ref, "passage", p_ids, p_mask, 32, True, ac_dtype=torch.float16)
    out["Q_ac16"] = encode_all(ref, "query", q_ids, q_mask, 16, True, ac_dtype=torch.float16)

    def yard(Pa, Qa):
        ok_p, ok_q = np.isfinite(Pa).all(1), np.isfinite(Qa).all(1)
        cos = torch.nn.functional.cosine_similarity(torch.from_numpy(out["P_f32"][ok_p]).double(), torch.from_numpy(Pa[ok_p]).double(), dim=1)
        dd = np.abs(Qa[ok_q].astype(np.float64) @ Pa[ok_p].astype(np.float64).T
                    - out["Q_f32"][ok_q].astype(np.float64) @ out["P_f32"][ok_p].astype(np.float64).T)
        return [float(cos.min()), float(cos.mean()), float(dd.max())], int((~ok_p).sum() + (~ok_q).sum())
    out["ac_vs_f32"] = np.array(yard(out["P_ac"], out["Q_ac"])[0])
    y16, bad16 = yard(out["P_ac16"], out["Q_ac16"])
    out["ac16_vs_f32"] = np.array(y16)
    out["ac16_nonfinite_rows"] = np.array(bad16)
    print("  GTR-base reference bf16 autocast vs fp32: min cos %.6f mean cos %.6f max|ddot| %.5f" % tuple(out["ac_vs_f32"]))
    print("  GTR-base reference float16 autocast vs fp32: min cos %.8f mean cos %.8f max|ddot| %.6f  (%d non-finite rows)" % (*y16, bad16))
    np.savez_compressed(os.path.join(OUT, "gtr_base.npz"), **out)
    print("wrote gtr_base.npz")


def bert_large_rr(rng):
    torch.manual_seed(2)
    cfg = BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    lm = BertModel(cfg).eval()
    head = LinearHead(1024, 1)
    ref = RRModel(lm=lm, head=head, pooling="first", model_args=NS(encoder_only=False)).eval()
    n, L = 32, 162
    ids, mask = mg.synth_batch(rng, n, L, cfg.vocab_size, 40)
    # the reference's pair format (retriever/reranker.py:23-29, encode_plus(item1 + item2)): ONE sequence
    # [CLS] q d [SEP], token_type_ids all 0
    tt = np.zeros_like(ids)
    out = {"input_ids": ids.astype(np.uint16), "len": lengths(mask), "weight_checksum": checksum(lm),
           "head_w": head.linear.weight.detach().numpy()}
    # "ac16": float16 autocast, the reference's real `--fp16` mode (retriever/reranker.py:90-133 under torch.cuda.amp); "ac": bf16 autocast
    for tag, ac in (("f32", None), ("ac", torch.bfloat16), ("ac16", torch.float16)):
        sc = []
        t0 = time.time()
        for s in range(0, n, 8):
            batch = {"input_ids": torch.from_numpy(ids[s:s + 8]), "attention_mask": torch.from_numpy(mask[s:s + 8]),
                     "token_type_ids": torch.from_numpy(tt[s:s + 8])}
            with torch.no_grad():
                if ac is not None:
                    with torch.autocast("cpu", dtype=ac):
                        sc.append(ref.encode(batch).float().numpy())
                else:
                    sc.append(ref.encode(batch).float().numpy())
        out[f"scores_{tag}"] = np.concatenate(sc)[:, 0]
        print(f"    bert-large pairs autocast={ac}: {time.time() - t0:.1f} s", flush=True)
    out["ac_vs_f32"] = np.array([float(np.abs(out["scores_ac"] - out["scores_f32"]).max()), float(np.abs(out["scores_f32"]).max())])
    out["ac16_vs_f32"] = np.array([float(np.abs(out["scores_ac16"] - out["scores_f32"]).max()), float(np.abs(out["scores_f32"]).max())])
    print("  bert-large RR reference bf16 autocast vs fp32: max|dscore| %.5f (|score| <= %.3f)" % tuple(out["ac_vs_f32"]))
    print("  bert-large RR reference float16 autocast vs fp32: max|dscore| %.6f" % out["ac16_vs_f32"][0])
    np.savez_compressed(os.path.join(OUT, "bert_large_rr.npz"), **out)
    print("wrote bert_large_rr.npz")


TRAIN_STRIDE = 11


def train_subset_rows(name, shape, used_tokens):
    """Row indices of parameter `name` whose gradient the fixture keeps (None: the whole tensor)."""
    if name == "embeddings.word_embeddings.weight":
        return used_tokens[::3]
    if len(shape) == 2 and shape[0] * shape[1] > 200_000:
        return np.arange(0, shape[0], TRAIN_STRIDE)
    return None


def train_base(rng):
    from openmatch.modeling import DRModel  # the reference
    torch.manual_seed(3)
    cfg = BertConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    n_psg, nq = 8, 8
    model = DRModel(lm_q=lm, lm_p=lm, pooling="first", model_args=NS(encoder_only=False), data_args=NS(train_n_passages=n_psg),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=nq))
    model.train()
    q_ids, q_mask = mg.synth_batch(rng, nq, 32, cfg.vocab_size, 4)
    p_ids, p_mask = mg.synth_batch(rng, nq * n_psg, 128, cfg.vocab_size, 16)
    mk = lambda ids, mask: {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
                            "token_type_ids": torch.zeros_like(torch.from_numpy(ids))}
    q, p = mk(q_ids, q_mask), mk(p_ids, p_mask)
    used = np.unique(np.concatenate([q_ids[q_mask > 0], p_ids[p_mask > 0]]))
    params = [(k, v) for k, v in lm.named_parameters()]

    def run(ac_dtype, loss_scale, enc_only=False):
        """enc_only: autocast covers the two encoder calls only; scores and cross entropy (modeling/dense_retrieval_model.py:
        113-122, the reference's own `loss_fn` member) are evaluated in fp32 on the float()-ed representations."""
        for _, v in params:
            v.grad = None
        t0 = time.time()
        if ac_dtype is None:
            o = model(query=q, passage=p)
            loss, scores = o.loss, o.scores
        elif not enc_only:
            with torch.autocast("cpu", dtype=ac_dtype):
                o = model(query=q, passage=p)
            loss, scores = o.loss, o.scores
        else:
            with torch.autocast("cpu", dtype=ac_dtype):
                _, q_reps = model.encode_query(q)
                _, p_reps = model.encode_passage(p)
            scores = torch.matmul(q_reps.float(), p_reps.float().transpose(0, 1))
            target = torch.arange(scores.size(0), dtype=torch.long) * n_psg
            loss = model.loss_fn(scores, target)
        (loss.float() * loss_scale).backward()
        grads = {k: (v.grad.detach().double() / loss_scale) for k, v in params if v.grad is not None}
        assert all(torch.isfinite(g_).all() for g_ in grads.values()), "non-finite gradient: lower the loss scale"
        print(f"    train step autocast={ac_dtype} enc_only={enc_only}: loss {float(loss):.6f}  {time.time() - t0:.1f} s", flush=True)
        return float(loss), scores.detach().float().numpy(), grads

    loss32, scores32, g32 = run(None, 1.0)
    loss_bf, _, gbf = run(torch.bfloat16, 1.0)
    # float16: a static scale of 2^12 stands in for GradScaler's dynamic one (same arithmetic: the loss is multiplied before
    # backward, the f32 parameter gradients divided afterwards; a power of two changes no mantissa)
    loss_h, _, gh = run(torch.float16, 4096.0)
    # The whole-forward autocast runs are LOOSE yardsticks on a random-init model: every CLS dot is ~762 +- 0.3 and autocast
    # rounds the score matrix itself to 16 bits (a bf16 ulp at 762 is 4).  The tight ones keep the loss in fp32:
    loss_bf_e, _, gbf_e = run(torch.bfloat16, 1.0, enc_only=True)
    loss_h_e, _, gh_e = run(torch.float16, 4096.0, enc_only=True)
    out = {"q_input_ids": q_ids.astype(np.uint16), "q_len": lengths(q_mask), "p_input_ids": p_ids.astype(np.uint16),
           "p_len": lengths(p_mask), "weight_checksum": checksum(lm), "n_psg": np.array(n_psg),
           "loss_f32": np.array(loss32), "loss_acbf16": np.array(loss_bf), "loss_ac16": np.array(loss_h), "scores_f32": scores32,
           "loss_acbf16_enc": np.array(loss_bf_e), "loss_ac16_enc": np.array(loss_h_e),
           "stride": np.array(TRAIN_STRIDE)}
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
    names, yard = [], []
    for k, v in params:
        if k not in g32:
            continue
        rows = train_subset_rows(k, tuple(v.shape), used)
        full = g32[k]
        sub = (lambda t: t) if rows is None else (lambda t, r=torch.from_numpy(rows): t[r])
        out["g::" + k] = sub(full).float().numpy()
        if rows is not None:
            out["rows::" + k] = rows.astype(np.int32)
        names.append(k)
        # columns: bf16-autocast on the subset, on the whole tensor; float16-autocast on the subset, on the whole tensor; |g|_2 (whole);
        # then the same four for the encoder-only autocast runs (fp32 loss); |g|_2 of the subset
        yard.append([rel(sub(gbf[k]), sub(full)), rel(gbf[k], full), rel(sub(gh[k]), sub(full)), rel(gh[k], full), float(full.norm()),
                     rel(sub(gbf_e[k]), sub(full)), rel(gbf_e[k], full), rel(sub(gh_e[k]), sub(full)), rel(gh_e[k], full), float(sub(full).norm())])
    out["grad_names"] = np.array(names)
    out["yardstick"] = np.array(yard)
    y = out["yardstick"]
    print("  train_base: %d gradient tensors; reference bf16-autocast rel-L2 from fp32: median %.2e max %.2e; float16-autocast: median %.2e max %.2e"
          % (len(names), np.median(y[:, 1]), y[:, 1].max(), np.median(y[:, 3]), y[:, 3].max()))
    print("              encoder-only autocast (fp32 loss): bf16 median %.2e max %.2e; float16 median %.2e max %.2e"
          % (np.median(y[:, 6]), y[:, 6].max(), np.median(y[:, 8]), y[:, 8].max()))
    for n_, r_ in zip(names, y):
        print("    %-52s whole fwd: bf16 %.2e f16 %.2e | encoder only: bf16 %.2e (sub %.2e) f16 %.2e (sub %.2e)  |g| %.2e"
              % (n_, r_[1], r_[3], r_[6], r_[5], r_[8], r_[7], r_[4]))
    np.savez_compressed(os.path.join(OUT, "train_base.npz"), **out)
    print("wrote train_base.npz  (%.1f MB)" % (os.path.getsize(os.path.join(OUT, "train_base.npz")) / 1e6))


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    what = sys.argv[1:] or ["config1", "gtr", "large"]
    if "train" in what:
        train_base(np.random.default_rng(SEED + 15))
    if "config1" in what:
        config1(np.random.default_rng(SEED + 11))
    if "spread" in what:
        config1_spread(np.random.default_rng(SEED + 14))
    if "spread_qrels" in what:
        spread_qrels()
    if "config1_ac16" in what:
        config1_ac16()
    if "gtr" in what:
        gtr_base(np.random.default_rng(SEED + 12))
    if "large" in what:
        bert_large_rr(np.random.default_rng(SEED + 13))


if __name__ == "__main__":
    main()
