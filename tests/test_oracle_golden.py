"""CPU: the oracle restatements against the fixtures produced by running the REFERENCE
(oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import encoder_ref, flatip, retrieval_ref
from tests.helpers import items_from_golden, model_from_golden

CASES = [("bert_tiny_first", "bert", False), ("bert_tiny_mean_head_norm", "bert", False),
         ("t5_tiny_gtr", "t5", False), ("t5_tiny_gated", "t5", True)]


@pytest.mark.parametrize("name,arch,gated", CASES)
def test_encoder_oracle_matches_reference(golden, name, arch, gated):
    g = golden(name)
    cfg, model = model_from_golden(g, arch, gated)
    pooling, has_head, normalize, _ = g["meta"]
    head_w = torch.from_numpy(g["head_w"]) if has_head == "1" else None
    for kind in ("p", "q"):
        hidden, reps = encoder_ref.encode(model.state_dict(), cfg, arch, items_from_golden(g, kind),
                                          str(pooling), head_w, normalize == "1")
        assert np.abs(reps.numpy() - g[kind + "_reps"]).max() < 2e-5
        if kind == "p":
            assert np.abs(hidden[:3].numpy() - g["p_hidden"]).max() < 2e-5
    # and against HF itself, live (transformers ships in the image)
    with torch.no_grad():
        it = items_from_golden(g, "q")
        hf = model(**it, return_dict=True).last_hidden_state
        mine = encoder_ref.encode(model.state_dict(), cfg, arch, it, "first")[0]
    assert (hf - mine).abs().max() < 2e-5


def test_bert_base_oracle_matches_reference(golden):
    from transformers import BertConfig, BertModel
    g = golden("bert_base_seed0")
    torch.manual_seed(0)
    cfg = BertConfig()
    model = BertModel(cfg).eval()
    sd = model.state_dict()
    chk = np.array([float(sum(v.double().sum() for v in sd.values())),
                    float(sum(v.double().abs().sum() for v in sd.values()))])
    assert np.allclose(chk, g["weight_checksum"], rtol=1e-12), "seeded bert-base weights differ from the fixture's"
    _, reps = encoder_ref.encode(sd, cfg, "bert", items_from_golden(g, "q"), "first")
    assert np.abs(reps.numpy() - g["q_reps"]).max() < 2e-5


def test_flatip_and_retrieval_plumbing(golden):
    g = golden("retrieval_1k")
    idx = flatip.IndexFlatIP(768)
    idx.add(g["P"][:500]); idx.add(g["P"][500:])
    assert idx.ntotal == 1000
    D, I = idx.search(g["Q"], 100)
    assert (I == g["I"]).all() and np.abs(D - g["D"]).max() == 0
    assert (np.diff(D, axis=1) <= 0).all()
    D64, I64 = idx.search(g["Q"], 100, dtype=torch.float64)
    assert (np.sort(I64, 1) == np.sort(I, 1)).mean() > 0.99   # fp32 vs fp64 differ only at boundary near-ties
    # padding semantics when ntotal < k
    small = flatip.IndexFlatIP(768); small.add(g["P"][:7])
    D7, I7 = small.search(g["Q"][:3], 10)
    assert (I7[:, 7:] == -1).all() and (D7[:, 7:] == np.float32(-3.4028235e38)).all()
    run = retrieval_ref.search_to_dict(D, I, list(g["doc_ids"]), list(g["qry_ids"]))
    assert "".join(retrieval_ref.trec_lines(run)) == str(g["trec"])
    qrel = {q: {d: 1} for q, d in zip(g["qry_ids"], g["qrel_docs"])}
    assert retrieval_ref.eval_mrr(qrel, run, cutoff=10)["all"] == pytest.approx(float(g["mrr10"]), abs=0)
    halves = [{q: dict(list(h.items())[:60]) for q, h in run.items()},
              {q: dict(list(h.items())[40:]) for q, h in run.items()}]
    merged = retrieval_ref.merge_retrieval_results_by_score(halves, 50)
    assert [" ".join(merged[q]) for q in g["qry_ids"]] == list(g["merged_keys"])
    idx.reset(); assert idx.ntotal == 0


def test_t5_bucket_oracle(golden):
    g = golden("t5_buckets")
    assert (encoder_ref.t5_relative_bucket(torch.from_numpy(g["rel"])).numpy() == g["bucket"]).all()


def test_contrastive_oracle(golden):
    g = golden("train_bert_tiny")
    loss, scores = retrieval_ref.contrastive_loss(torch.from_numpy(g["q_reps"]), torch.from_numpy(g["p_reps"]),
                                                  int(g["n_psg"]))
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    assert np.abs(scores.numpy() - g["scores"]).max() < 1e-6


def test_training_oracle_at_bert_base_width_matches_reference_gradients(golden):
    """tests/golden/train_base.npz (the reference's DRModel.forward + backward at the benchmarked training shape): the CPU oracle's
    autograd reproduces its fp32 loss and gradients; the autocast yardsticks in it are ordered as 16-bit formats must be."""
    from transformers import BertConfig, BertModel
    g = golden("train_base")
    torch.manual_seed(3)
    cfg = BertConfig(num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    lm = BertModel(cfg)
    sd = lm.state_dict()
    chk = np.array([float(sum(v.double().sum() for v in sd.values())), float(sum(v.double().abs().sum() for v in sd.values()))])
    assert np.allclose(chk, g["weight_checksum"], rtol=1e-12)

    def items(prefix, L):
        ids = torch.from_numpy(g[prefix + "input_ids"].astype(np.int64))
        lens = torch.from_numpy(g[prefix + "len"].astype(np.int64))
        return {"input_ids": ids, "attention_mask": (torch.arange(L)[None, :] < lens[:, None]).long()}
    params = dict(lm.named_parameters())
    params.update(dict(lm.named_buffers()))
    q = encoder_ref.encode(params, cfg, "bert", items("q_", 32), "first")[1]
    p = encoder_ref.encode(params, cfg, "bert", items("p_", 128), "first")[1]
    loss, scores = retrieval_ref.contrastive_loss(q, p, int(g["n_psg"]))
    assert abs(float(loss.detach()) - float(g["loss_f32"])) < 1e-4          # softmax over dots of ~762 (one f32 ulp = 6e-5)
    assert np.abs(scores.detach().numpy() - g["scores_f32"]).max() < 2e-3          # dots of ~762: a few f32 ulps
    loss.backward()
    named = dict(lm.named_parameters())
    y = g["yardstick"]
    for i, name in enumerate(str(n) for n in g["grad_names"]):
        got = named[name].grad.double()
        if "rows::" + name in g.files:
            got = got[torch.from_numpy(g["rows::" + name].astype(np.int64))]
        ref = torch.from_numpy(g["g::" + name]).double()
        err = float((got - ref).norm())
        assert err < 2e-3 * float(ref.norm()) or err < 1e-6, (name, err, float(ref.norm()))
        if y[i, 4] > 1e-6:       # float16 autocast is closer to fp32 than bfloat16 autocast; keeping the loss in fp32 helps both
            assert y[i, 3] < y[i, 1] and y[i, 8] < y[i, 6] and y[i, 6] < y[i, 1], (name, y[i])
