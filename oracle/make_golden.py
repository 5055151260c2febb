"""Generates tests/golden/*.npz by EXECUTING THE REFERENCE in the build container, and pins the
oracle restatements against it.  Run:  python oracle/make_golden.py   (needs /root/reference).

What runs here is the reference's own code — `openmatch.modeling.DRModel(ForInference)`,
`openmatch.loss`, `openmatch.utils`, `openmatch.retriever.Retriever.search`, `eval_mrr` lifted
verbatim (by AST, at run time) out of scripts/evaluate.py — over HF transformers' BertModel /
T5EncoderModel with seeded random weights (no pretrained checkpoints exist offline).  `faiss` is
absent, so `oracle.flatip` is injected as `sys.modules['faiss']` for the retriever import.
The fixtures carry inputs, weights (tiny models) and the reference's outputs; the GPU tests
compare the HIP path with them without needing /root/reference.
"""
import ast
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, os.path.join(REF, "src"))   # `openmatch` == the reference
sys.path.insert(1, REPO)                        # `oracle`

import numpy as np  # noqa: E402
import torch  # noqa: E402
from transformers import BertConfig, BertModel, T5Config, T5EncoderModel  # noqa: E402

from oracle import encoder_ref, flatip, retrieval_ref  # noqa: E402

import datasets  # noqa: E402,F401  (probes for a real faiss at import; must come before the stub)

faiss_stub = types.ModuleType("faiss")
faiss_stub.IndexFlatIP = flatip.IndexFlatIP
sys.modules["faiss"] = faiss_stub

import openmatch  # noqa: E402
assert openmatch.__file__.startswith(REF), openmatch.__file__
from openmatch.modeling import DRModel, DRModelForInference  # noqa: E402
from openmatch.modeling.linear import LinearHead  # noqa: E402
from openmatch.retriever import Retriever  # noqa: E402
from openmatch.utils import merge_retrieval_results_by_score, save_as_trec  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
SEED = 20260925
NS = types.SimpleNamespace


def reference_eval_mrr():
    src = open(os.path.join(REF, "scripts", "evaluate.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "eval_mrr"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "evaluate.py", "exec"), ns)
    return ns["eval_mrr"]


def synth_batch(rng, n, L, vocab, lo_len, bert=True):
    """MS-MARCO-shaped ids: [CLS] ... [SEP] pad (BERT) / ... </s> pad (T5), random real length."""
    ids = np.zeros((n, L), np.int64)
    mask = np.zeros((n, L), np.int64)
    for i in range(n):
        ln = int(rng.integers(lo_len, L + 1))
        body = rng.integers(min(1000, vocab // 2), vocab, size=ln)
        if bert:
            body[0], body[-1] = 101 % vocab, 102 % vocab
        else:
            body[-1] = 1
        ids[i, :ln] = body
        mask[i, :ln] = 1
    return ids, mask


def sd_np(model):
    return {k: v.detach().numpy() for k, v in model.state_dict().items()}


def check(name, a, b, tol):
    d = float((torch.as_tensor(a, dtype=torch.float64) - torch.as_tensor(b, dtype=torch.float64)).abs().max())
    print(f"  pin {name:58s} max|oracle-reference| = {d:.3e}")
    assert d <= tol, (name, d)


def encoder_case(tag, arch, cfg, model, pooling, head, normalize, n_psg, n_qry, store_weights, rng):
    model.eval()
    margs = NS(encoder_only=(arch == "t5"))
    ref = DRModelForInference(lm_q=model, lm_p=model, pooling=pooling, head_q=head, head_p=head,
                              normalize=normalize, model_args=margs)
    vocab = cfg.vocab_size
    p_ids, p_mask = synth_batch(rng, n_psg, 128, vocab, 16, arch == "bert")
    q_ids, q_mask = synth_batch(rng, n_qry, 32, vocab, 4, arch == "bert")
    out = {}
    for kind, ids, mask in (("p", p_ids, p_mask), ("q", q_ids, q_mask)):
        items = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        if arch == "bert":
            tt = np.zeros_like(ids)
            tt[:, ids.shape[1] // 2:] = (mask[:, ids.shape[1] // 2:] > 0)   # exercise the type table
            items["token_type_ids"] = torch.from_numpy(tt)
            out[kind + "_token_type_ids"] = tt
        with torch.no_grad():
            hidden, reps = ref.encode(items, model, head)
        o_hidden, o_reps = encoder_ref.encode(model.state_dict(), cfg, arch, items, pooling,
                                              None if head is None else head.linear.weight, normalize)
        check(f"{tag} {kind} hidden", o_hidden, hidden, 2e-5)
        check(f"{tag} {kind} reps", o_reps, reps, 2e-5)
        out[kind + "_input_ids"], out[kind + "_attention_mask"] = ids, mask
        out[kind + "_reps"] = reps.numpy()
        if store_weights and kind == "p":
            out[kind + "_hidden"] = hidden[:3].numpy()      # first three passages only (fixture size)
    with torch.no_grad():
        out["scores"] = (torch.from_numpy(out["q_reps"]) @ torch.from_numpy(out["p_reps"]).t()).numpy()
    if store_weights:
        for k, v in sd_np(model).items():
            out["w::" + k] = v
        if head is not None:
            out["head_w"] = head.linear.weight.detach().numpy()
    else:
        sd = model.state_dict()
        out["weight_checksum"] = np.array([float(sum(v.double().sum() for v in sd.values())),
                                           float(sum(v.double().abs().sum() for v in sd.values()))])
    out["meta"] = np.array([pooling, str(int(head is not None)), str(int(normalize)), arch])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
    print(f"wrote {tag}.npz")
    return out


def training_case(rng):
    """Reference DRModel.forward in training mode (dropout 0) on a tiny BERT: loss, scores, grads."""
    torch.manual_seed(SEED + 5)
    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     vocab_size=600, max_position_embeddings=160, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    head = LinearHead(128, 128)
    n_psg = 2
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True,
                    model_args=NS(encoder_only=False), data_args=NS(train_n_passages=n_psg),
                    train_args=NS(negatives_x_device=False, per_device_train_batch_size=4))
    model.train()
    q_ids, q_mask = synth_batch(rng, 4, 32, cfg.vocab_size, 4)
    p_ids, p_mask = synth_batch(rng, 8, 128, cfg.vocab_size, 16)
    q = {"input_ids": torch.from_numpy(q_ids), "attention_mask": torch.from_numpy(q_mask),
         "token_type_ids": torch.zeros_like(torch.from_numpy(q_ids))}
    p = {"input_ids": torch.from_numpy(p_ids), "attention_mask": torch.from_numpy(p_mask),
         "token_type_ids": torch.zeros_like(torch.from_numpy(p_ids))}
    o = model(query=q, passage=p)
    o.loss.backward()
    l2, s2 = retrieval_ref.contrastive_loss(o.q_reps.detach(), o.p_reps.detach(), n_psg)
    check("train loss", l2, o.loss.detach(), 1e-6)
    out = {"q_input_ids": q_ids, "q_attention_mask": q_mask, "p_input_ids": p_ids, "p_attention_mask": p_mask,
           "loss": o.loss.detach().numpy(), "scores": o.scores.detach().numpy(),
           "q_reps": o.q_reps.detach().numpy(), "p_reps": o.p_reps.detach().numpy(),
           "head_w": head.linear.weight.detach().numpy(), "g::head_w": head.linear.weight.grad.numpy(),
           "n_psg": np.array(n_psg)}
    for k, v in lm.state_dict().items():
        out["w::" + k] = v.detach().numpy()
    for k, v in lm.named_parameters():
        if v.grad is not None:
            out["g::" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "train_bert_tiny.npz"), **out)
    print("wrote train_bert_tiny.npz")


def retrieval_case(rng):
    """Reference Retriever.search (+ injected FlatIP), save_as_trec, merge, eval_mrr on a
    1k-passage / 100-query embedding set (BASELINE config 1 plumbing)."""
    import pickle
    import tempfile
    d, n, nq, k = 768, 1000, 100, 100
    mean = rng.standard_normal(d).astype(np.float32)
    P = (mean + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    Q = (mean + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
    P /= np.linalg.norm(P, axis=1, keepdims=True)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    doc_ids = [f"D{7 * i + 3}" for i in range(n)]
    qry_ids = [f"Q{i}" for i in range(nq)]
    with tempfile.TemporaryDirectory() as tmp:
        # two corpus shards + two query shards, exactly as two encoding ranks would leave them
        for r in range(2):
            with open(os.path.join(tmp, f"embeddings.corpus.rank.{r}"), "wb") as f:
                pickle.dump((P[r * 500:(r + 1) * 500], doc_ids[r * 500:(r + 1) * 500]), f, protocol=4)
            with open(os.path.join(tmp, f"embeddings.query.rank.{r}"), "wb") as f:
                pickle.dump((Q[r * 50:(r + 1) * 50], qry_ids[r * 50:(r + 1) * 50]), f, protocol=4)
        args = NS(device="cpu", output_dir=tmp, world_size=2, process_index=0, local_process_index=0)
        dummy = torch.nn.Linear(1, 1)
        r = Retriever(dummy, None, args)
        # glob order is OS dependent; feed partitions explicitly in rank order
        for part in sorted(os.listdir(tmp)):
            if part.startswith("embeddings.corpus"):
                r.init_index_and_add(os.path.join(tmp, part)) if r.doc_lookup == [] else None
        r.reset_index()
        r.index = flatip.IndexFlatIP(d)
        for rk in range(2):
            with open(os.path.join(tmp, f"embeddings.corpus.rank.{rk}"), "rb") as f:
                e, ids = pickle.load(f)
            r.index.add(e)
            r.doc_lookup.extend(ids)
        run = r.search(k)
        trec = os.path.join(tmp, "run.trec")
        save_as_trec(run, trec)
        trec_text = open(trec).read()
    # qrels: one relevant doc per query at oracle rank r ~ U{1..20}
    qrel = {}
    for qid in qry_ids:
        ranked = sorted(run[qid].items(), key=lambda kv: kv[1], reverse=True)
        qrel[qid] = {ranked[int(rng.integers(0, 20))][0]: 1}
    mrr = reference_eval_mrr()(qrel, run, cutoff=10)
    check("eval_mrr restatement", retrieval_ref.eval_mrr(qrel, run, cutoff=10)["all"], mrr["all"], 0)
    assert "".join(retrieval_ref.trec_lines(run)) == trec_text
    halves = [{q: dict(list(h.items())[:60]) for q, h in run.items()}, {q: dict(list(h.items())[40:]) for q, h in run.items()}]
    merged = merge_retrieval_results_by_score(halves, 50)
    assert merged == retrieval_ref.merge_retrieval_results_by_score(halves, 50)
    I = np.array([[doc_ids.index(dn) for dn in run[q]] for q in qry_ids], np.int64)
    D = np.array([[run[q][dn] for dn in run[q]] for q in qry_ids], np.float32)
    np.savez_compressed(os.path.join(OUT, "retrieval_1k.npz"), P=P, Q=Q, doc_ids=np.array(doc_ids),
                        qry_ids=np.array(qry_ids), I=I, D=D, trec=np.array(trec_text),
                        qrel_docs=np.array([list(qrel[q])[0] for q in qry_ids]), mrr10=np.array(mrr["all"]),
                        merged_keys=np.array([" ".join(merged[q]) for q in qry_ids]))
    print("wrote retrieval_1k.npz   MRR@10 =", mrr["all"])


def bucket_case():
    from transformers.models.t5.modeling_t5 import T5Attention
    rel = torch.arange(-600, 601)
    b = T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=32, max_distance=128)
    check("t5 bucket restatement", encoder_ref.t5_relative_bucket(rel), b, 0)
    np.savez_compressed(os.path.join(OUT, "t5_buckets.npz"), rel=rel.numpy(), bucket=b.numpy())
    print("wrote t5_buckets.npz")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(SEED)
    torch.manual_seed(SEED)
    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=600,
                     max_position_embeddings=160)
    encoder_case("bert_tiny_first", "bert", cfg, BertModel(cfg), "first", None, False, 6, 4, True, rng)
    torch.manual_seed(SEED + 1)
    encoder_case("bert_tiny_mean_head_norm", "bert", cfg, BertModel(cfg), "mean", LinearHead(128, 128), True, 6, 4, True, rng)
    torch.manual_seed(SEED + 2)
    tcfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_heads=2, d_kv=64, feed_forward_proj="relu", vocab_size=600)
    encoder_case("t5_tiny_gtr", "t5", tcfg, T5EncoderModel(tcfg), "mean", LinearHead(128, 128), True, 6, 4, True, rng)
    torch.manual_seed(SEED + 3)
    gcfg = T5Config(d_model=128, d_ff=256, num_layers=2, num_heads=2, d_kv=64, feed_forward_proj="gated-gelu", vocab_size=600)
    encoder_case("t5_tiny_gated", "t5", gcfg, T5EncoderModel(gcfg), "mean", None, False, 6, 4, True, rng)
    # bert-base sized: weights are re-created from the seed by the tests (HF is in the image)
    torch.manual_seed(0)
    bcfg = BertConfig()
    encoder_case("bert_base_seed0", "bert", bcfg, BertModel(bcfg), "first", None, False, 8, 4, False, rng)
    training_case(rng)
    retrieval_case(rng)
    bucket_case()


if __name__ == "__main__":
    main()
