"""CPU: host-side logic of the reference-shaped API (no kernels run here)."""
import os
import pickle

import numpy as np
import pytest
import torch

from tests.conftest import REPO
from tests.helpers import NS, tiny_bert_config


def test_trec_roundtrip_templates_and_merge(tmp_path, golden):
    from openmatch.utils import (eval_mrr, fill_template, find_all_markers, load_from_trec,
                                 merge_retrieval_results_by_score, save_as_trec)
    g = golden("retrieval_1k")
    run = {q: {g["doc_ids"][i]: float(s) for i, s in zip(g["I"][j], g["D"][j])} for j, q in enumerate(g["qry_ids"])}
    path = tmp_path / "run.trec"
    save_as_trec(run, str(path))
    assert path.read_text() == str(g["trec"])                      # byte-identical to the reference's writer
    back = load_from_trec(str(path))
    assert back.keys() == run.keys() and all(list(back[q]) == list(run[q]) for q in run)
    assert len(load_from_trec(str(path), max_len_per_q=7)["Q3"]) == 7
    assert load_from_trec(str(path), as_list=True)["Q0"][0][0] == list(run["Q0"])[0]
    qrel = {q: {d: 1} for q, d in zip(g["qry_ids"], g["qrel_docs"])}
    assert eval_mrr(qrel, back, cutoff=10)["all"] == pytest.approx(float(g["mrr10"]), abs=1e-12)
    halves = [{q: dict(list(h.items())[:60]) for q, h in run.items()}, {q: dict(list(h.items())[40:]) for q, h in run.items()}]
    merged = merge_retrieval_results_by_score(halves, 50)
    assert [" ".join(merged[q]) for q in g["qry_ids"]] == list(g["merged_keys"])
    assert find_all_markers("Title: <title> Text: <text>") == ["title", "text"]
    assert fill_template("<a.b>-<c>", {"a": {"b": 1}, "c": "x"}) == "1-x"
    with pytest.raises(ValueError, match="Cannot find the marker"):
        fill_template("<zz>", {})
    with pytest.warns(RuntimeWarning):
        assert fill_template("[<zz>]", {}, allow_not_found=True) == "[]"
    (tmp_path / "bad").write_text("a b\n")
    with pytest.raises(ValueError, match="Invalid run format"):
        load_from_trec(str(tmp_path / "bad"))


def test_argument_dataclasses_keep_reference_names_and_defaults():
    from dataclasses import fields
    from transformers import HfArgumentParser
    from openmatch.arguments import DataArguments, DRTrainingArguments, InferenceArguments, ModelArguments
    d = {f.name: f.default for f in fields(DataArguments)}
    assert (d["q_max_len"], d["p_max_len"], d["train_n_passages"], d["doc_template"]) == (32, 128, 8, "Title: <title> Text: <text>")
    m = {f.name: f.default for f in fields(ModelArguments)}
    assert (m["pooling"], m["feature"], m["normalize"], m["projection_in_dim"], m["encoder_only"]) == ("first", "last_hidden_state", False, 768, False)
    parser = HfArgumentParser((ModelArguments, DataArguments, InferenceArguments))
    ma, da, ia = parser.parse_args_into_dataclasses(["--model_name_or_path", "x", "--output_dir", "/tmp/o", "--use_gpu",
                                                     "--q_max_len", "16", "--pooling", "mean"])
    assert ma.pooling == "mean" and da.q_max_len == 16 and ia.use_gpu is True
    t = {f.name: f.default for f in fields(DRTrainingArguments)}
    assert t["warmup_ratio"] == 0.1 and t["negatives_x_device"] is False and t["gc_p_chunk_size"] == 32


def test_weight_packing_layout_matches_hf_module():
    """Packed device weights: q|k|v concatenated [3H,H], biases f32, matrices in the compute dtype."""
    from transformers import BertModel
    from openmatch_amd import native as N
    from openmatch_amd.encoder import packed_weights
    from openmatch.modeling import LinearHead
    lm = BertModel(tiny_bert_config()).eval()
    head = LinearHead(128, 64)
    pk = packed_weights(lm, head, N.OM_BF16, torch.device("cpu"))
    assert pk.cfg["hidden"] == 128 and pk.cfg["n_layers"] == 2 and pk.cfg["head_out"] == 64 and pk.cfg["act"] == N.ACT_GELU_ERF
    qkv = [t for t in pk.keep if tuple(t.shape) == (384, 128)]
    assert len(qkv) == 2 and qkv[0].dtype == torch.bfloat16
    at = lm.encoder.layer[0].attention.self
    want = torch.cat([at.query.weight, at.key.weight, at.value.weight]).to(torch.bfloat16)
    assert torch.equal(qkv[0], want)
    assert pk.layers[0].qkv_w == qkv[0].data_ptr()
    assert packed_weights(lm, head, N.OM_BF16, torch.device("cpu")) is pk           # cached
    with torch.no_grad():
        at.query.weight.add_(1.0)                                                  # in-place update bumps _version
    assert packed_weights(lm, head, N.OM_BF16, torch.device("cpu")) is not pk       # repacked after an optimizer step


def test_model_save_build_roundtrip_and_config(tmp_path):
    from transformers import BertModel
    from openmatch.modeling import DRModel, LinearHead
    lm = BertModel(tiny_bert_config())
    head = LinearHead(128, 128)
    model = DRModel(lm_q=lm, lm_p=lm, pooling="mean", head_q=head, head_p=head, normalize=True)
    out = tmp_path / "ckpt"; out.mkdir()
    model.save(str(out))
    assert sorted(os.listdir(out)) == sorted(["config.json", "model.safetensors", "linear.pt", "head_config.json", "openmatch_config.json"])
    cfg = model._get_config_dict()
    assert cfg == {"tied": True, "plm_backbone": {"type": "BertModel", "feature": "last_hidden_state"},
                   "pooling": "mean", "linear_head": True, "normalize": True}
    back = DRModel.build(NS(model_name_or_path=str(out), encoder_only=False, untie_encoder=False, add_linear_head=False,
                            feature="x", pooling="first", normalize=False, projection_in_dim=1, projection_out_dim=1))
    assert back.pooling == "mean" and back.normalize and back.tied and back.lm_q is back.lm_p
    assert torch.equal(back.head_q.linear.weight, head.linear.weight)
    assert torch.equal(back.lm_q.embeddings.word_embeddings.weight, lm.embeddings.word_embeddings.weight)
    # untied layout
    import copy
    m2 = DRModel(lm_q=lm, lm_p=copy.deepcopy(lm), tied=False)
    out2 = tmp_path / "untied"; out2.mkdir()
    m2.save(str(out2))
    assert {"query_model", "passage_model", "openmatch_config.json"} <= set(os.listdir(out2))
    with pytest.raises(ValueError, match="Distributed training has not been initialized"):
        DRModel(lm_q=lm, lm_p=lm, train_args=NS(negatives_x_device=True))


class _FakeEncoder(torch.nn.Module):
    """Stands in for the HIP encoder so the CPU suite can walk the Retriever's loops."""

    def forward(self, query=None, passage=None):
        from openmatch.modeling import DROutput
        x = query if query is not None else passage
        reps = torch.nn.functional.one_hot(x["input_ids"][:, 0] % 8, 8).float() + 0.01 * x["input_ids"][:, 1:2].float()
        return DROutput(q_reps=reps if query is not None else None, p_reps=reps if passage is not None else None)


def test_retriever_encoding_loop_writes_reference_pickles(tmp_path):
    from openmatch.retriever import Retriever
    rows = [{"text_id": f"d{i}", "input_ids": [i, i + 1, 0], "attention_mask": [1, 1, 0]} for i in range(10)]
    args = NS(device="cpu", output_dir=str(tmp_path), world_size=1, process_index=0, local_process_index=0, fp16=False,
              per_device_eval_batch_size=4, dataloader_num_workers=0, dataloader_pin_memory=False)
    r = Retriever(_FakeEncoder(), rows, args)
    r.doc_embedding_inference()
    with open(tmp_path / "embeddings.corpus.rank.0", "rb") as f:
        enc, ids = pickle.load(f)
    assert isinstance(enc, np.ndarray) and enc.dtype == np.float32 and enc.shape == (10, 8) and ids == [f"d{i}" for i in range(10)]
    r.query_embedding_inference(rows[:3])
    with open(tmp_path / "embeddings.query.rank.0", "rb") as f:
        qenc, qids = pickle.load(f)
    assert qenc.shape == (3, 8) and qids == ["d0", "d1", "d2"]
    with pytest.raises(ValueError, match="No corpus dataset provided"):
        Retriever(_FakeEncoder(), None, args).doc_embedding_inference()
    with pytest.raises(ValueError, match="Index is not initialized"):
        Retriever(_FakeEncoder(), None, args).search(5)
    # -1 padding maps to the LAST doc, as numpy fancy indexing does in the reference
    r.doc_lookup, r.query_lookup = ids, ["q"]
    hits = r._hits_to_dict(np.array([[0.5, -3.4e38]], np.float32), np.array([[2, -1]]), 2)
    assert list(hits["q"]) == ["d2", "d9"]


def test_trainer_schedule_param_groups_and_sharded_loader():
    from openmatch_amd.trainer.dense_trainer import DRTrainer, linear_schedule_factor, parameter_groups, split_dense_inputs
    from transformers import BertModel
    assert [linear_schedule_factor(s, 10, 100) for s in (0, 5, 10, 100)] == [0.0, 0.5, 1.0, 0.0]
    lm = BertModel(tiny_bert_config())
    decay, no_decay = parameter_groups(lm, 0.01)
    assert decay["weight_decay"] == 0.01 and no_decay["weight_decay"] == 0.0
    n_all = sum(1 for _ in lm.parameters())
    assert len(decay["params"]) + len(no_decay["params"]) == n_all
    assert all(p.dim() == 1 for p in no_decay["params"])

    class Stream(torch.utils.data.IterableDataset):
        def __iter__(self):
            return iter(range(16))
    seen = []
    for rank in range(2):
        args = NS(world_size=2, process_index=rank, per_device_train_batch_size=2, dataloader_num_workers=0,
                  dataloader_pin_memory=False, negatives_x_device=False)
        t = DRTrainer(model=None, args=args, train_dataset=Stream(), data_collator=lambda b: b)
        seen.append([x for b in t.get_train_dataloader() for x in b])
    assert seen[0] == [0, 1, 4, 5, 8, 9, 12, 13] and seen[1] == [2, 3, 6, 7, 10, 11, 14, 15]   # reference interleave
    chunks = split_dense_inputs({"query": {"input_ids": torch.arange(10).view(5, 2), "attention_mask": torch.ones(5, 2)}}, 2)
    assert [c["query"]["input_ids"].shape[0] for c in chunks] == [2, 2, 1]


def test_trainer_epoch_is_advanced_before_each_epoch_iterator(tmp_path):
    """TrainDataset.__iter__ reads int(trainer.state.epoch) when an epoch's iterator is created
    (reference train_dataset.py:115-119); HF Trainer has advanced it by then.  With max_steps set, the k-th
    pass over the data must see epoch k (round-1 bug: passes 0 and 1 both saw epoch 0)."""
    from openmatch.trainer import DRTrainer

    class Data(torch.utils.data.IterableDataset):
        def __init__(self):
            self.trainer, self.seen = None, []

        def __iter__(self):
            self.seen.append(int(self.trainer.state.epoch))
            for i in range(4):
                yield i

        def __len__(self):
            return 4

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(1))

        def forward(self, query=None, passage=None):
            return NS(loss=(self.w * query.float().mean()).sum())

    data = Data()
    args = NS(device="cpu", per_device_train_batch_size=2, max_steps=5, learning_rate=1e-3, output_dir=str(tmp_path),
              num_train_epochs=1, world_size=1, process_index=0, dataloader_pin_memory=False, logging_steps=1)
    trainer = DRTrainer(model=Model(), args=args, train_dataset=data,
                        data_collator=lambda items: (torch.tensor(items), torch.tensor(items)))
    data.trainer = trainer
    trainer.train()
    assert data.seen == [0, 1, 2]                      # 2 optimizer steps per pass, 5 steps
    assert [round(e["epoch"], 2) for e in trainer.state.log_history] == [0.5, 1.0, 1.5, 2.0, 2.5]


def test_no_kernel_spills_to_scratch():
    """Accumulators in scratch cost 5-7x (seen once with a lambda capture): keep every kernel of the
    library at private_segment_fixed_size == 0, except the two L<=256 attention variants."""
    import re
    import subprocess
    from openmatch_amd import native as N
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not available")
    # the device code sits in the library as a clang offload bundle: magic, count, then
    # (offset, size, triple-size, triple) records relative to the magic
    import struct
    blob = open(N.LIB_PATH, "rb").read()
    os.makedirs(os.path.join(REPO, "build"), exist_ok=True)
    kernels, base = {}, blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    while base >= 0:                                   # one bundle per translation unit
        (count,) = struct.unpack_from("<Q", blob, base + 24)
        pos = base + 32
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size:
                co = os.path.join(REPO, "build", "device_gfx950.co")
                open(co, "wb").write(blob[base + off:base + off + size])
                notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
                syms = re.findall(r"\.symbol:\s+(\S+)\.kd", notes)
                scratch = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)
                assert len(syms) == len(scratch)
                kernels.update(zip(syms, map(int, scratch)))
        base = blob.find(b"__CLANG_OFFLOAD_BUNDLE__", base + 1)
    assert len(kernels) > 60, len(kernels)
    # known exceptions: the L > 128 attention instantiations (KT = 6, 8) hold up to 128 score registers
    # and the one-wave-per-SIMD GEMM (gemm_nt_kernel6, 256 accumulators + 256 VGPRs per lane), whose
    # EPILOGUE may park a few values (checked in the ISA: nothing inside the K loop): <= 64 B for the
    # inference variants, <= 1.25 KiB for the training ones (dropout hash / pre-activation copy)
    def allowed(k, v):
        if ("attention_kernelI" in k or "attention_bwd_kernelI" in k) and ("Li8E" in k or "Li6E" in k):
            return True
        if "gemm_nt_kernel7" in k or "sim_filter_kernel7" in k or "attention_fwd16_kernel" in k:
            return False                 # generation 7 and the bf16 inference attention: no scratch at all
        if "gemm_nt_kernel6" in k:
            return v <= (1280 if re.search(r"Li\dELb1ELb[01]ELi\dEE", k) else 96)
        return False
    bad = {k: v for k, v in kernels.items() if v > 0 and not allowed(k, v)}
    assert not bad, bad


def test_packed_token_transport_round_trip():
    """openmatch_amd/feed.py: the collators' compact wire format (16-bit ids, one length per right-padded row, no token
    types when they are all zero) rebuilds exactly the int64 tensors the reference's collators would have sent
    (dataset/data_collator.py:78-91), and falls back to the plain batch whenever something does not fit."""
    import torch
    from openmatch_amd.feed import is_packed, pack_token_batch, packed_nbytes, unpack_token_batch
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 65536, (7, 33), generator=g)
    lens = torch.randint(1, 34, (7,), generator=g)
    mask = (torch.arange(33)[None, :] < lens[:, None]).long()
    for tti in (None, torch.zeros_like(ids), (torch.arange(33)[None, :] >= 10).long().expand(7, 33).contiguous()):
        batch = {"input_ids": ids, "attention_mask": mask}
        if tti is not None:
            batch["token_type_ids"] = tti
        packed = pack_token_batch(dict(batch))
        assert is_packed(packed)
        assert packed_nbytes(packed) * 4 < sum(v.numel() * 8 for v in batch.values())     # > 4x smaller (12x at [B,128] with no types)
        back = unpack_token_batch(packed, "cpu")
        assert back.keys() == batch.keys()
        for k in batch:
            assert back[k].dtype == torch.int64 and torch.equal(back[k], batch[k]), k
    # a mask with a hole travels as bytes; ids beyond 16 bits, negative ids or extra keys are not packed at all
    holed = mask.clone(); holed[0, 0] = 0
    back = unpack_token_batch(pack_token_batch({"input_ids": ids, "attention_mask": holed}), "cpu")
    assert torch.equal(back["attention_mask"], holed)
    assert not is_packed(pack_token_batch({"input_ids": ids + 70000, "attention_mask": mask}))
    neg = ids.clone(); neg[0, 0] = -1
    assert not is_packed(pack_token_batch({"input_ids": neg, "attention_mask": mask}))
    assert not is_packed(pack_token_batch({"input_ids": ids, "attention_mask": mask, "labels": ids}))
    plain = unpack_token_batch({"input_ids": ids, "attention_mask": mask}, "cpu")
    assert torch.equal(plain["input_ids"], ids)


def test_backbone_dispatch_and_position_offset():
    """openmatch_amd/encoder.py: BERT and RoBERTa-family modules map to the BERT stack (RoBERTa with its position table
    handed over from row padding_idx + 1, HF create_position_ids_from_input_ids), T5 to the T5 stack, anything else is
    refused by name -- the reference builds backbones with AutoModel (modeling/dense_retrieval_model.py:173)."""
    import pytest
    from transformers import BertConfig, BertModel, GPT2Config, GPT2Model, RobertaConfig, RobertaModel, T5Config, T5EncoderModel
    from openmatch_amd.encoder import _arch_of, position_offset
    tiny = dict(hidden_size=32, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64, vocab_size=50)
    bert, rob = BertModel(BertConfig(**tiny)), RobertaModel(RobertaConfig(max_position_embeddings=40, **tiny))
    t5 = T5EncoderModel(T5Config(d_model=64, d_ff=64, num_layers=1, num_heads=1, d_kv=64, vocab_size=50))
    assert _arch_of(bert) == "bert" and position_offset(bert) == 0
    assert _arch_of(rob) == "bert" and position_offset(rob) == rob.config.pad_token_id + 1 == 2
    assert _arch_of(t5) == "t5" and position_offset(t5) == 0
    with pytest.raises(NotImplementedError, match="GPT2Model"):
        _arch_of(GPT2Model(GPT2Config(n_embd=32, n_layer=1, n_head=1, vocab_size=50)))
    # RoBERTa: position t + offset equals HF's cumsum numbering only when no pad id precedes an attended token
    import torch
    from openmatch_amd.encoder import check_position_layout
    pad = rob.config.pad_token_id
    right = torch.tensor([[5, 6, 7, pad, pad], [5, 6, 7, 8, 9]]); rmask = (right != pad).long()
    check_position_layout(rob, right, rmask)
    hf_pos = (torch.cumsum(rmask, 1) * rmask + pad)[rmask.bool()]                        # create_position_ids_from_input_ids
    ours = (torch.arange(5).expand(2, 5) + position_offset(rob))[rmask.bool()]
    assert torch.equal(hf_pos, ours)
    left = torch.tensor([[pad, pad, 5, 6, 7]])
    with pytest.raises(ValueError, match="pad token precedes"):
        check_position_layout(rob, left, (left != pad).long())
    inside = torch.tensor([[5, pad, 6, 7, pad]])
    with pytest.raises(ValueError, match="pad token precedes"):
        check_position_layout(rob, inside, torch.tensor([[1, 1, 1, 1, 0]]))
    check_position_layout(bert, left, (left != pad).long())                               # BERT numbers positions 0..L-1 regardless


def test_one_pass_rule_for_tied_training_batches():
    """DRModel._one_pass_ok: queries ride along with the passages only for a TIED encoder in training mode with autograd
    on, equal key sets, queries no longer than passages and at most a quarter as many rows."""
    import torch
    from types import SimpleNamespace as NS
    from transformers import BertConfig, BertModel
    from openmatch_amd.modeling import DRModel
    lm = BertModel(BertConfig(hidden_size=32, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64, vocab_size=50))
    other = BertModel(lm.config)
    mk = lambda n, L, **extra: dict({"input_ids": torch.zeros(n, L, dtype=torch.long), "attention_mask": torch.ones(n, L, dtype=torch.long)}, **extra)
    args = dict(model_args=NS(encoder_only=False), data_args=NS(train_n_passages=8), train_args=NS(negatives_x_device=False))
    tied = DRModel(lm_q=lm, lm_p=lm, **args).train()
    assert tied._one_pass_ok(mk(8, 32), mk(64, 128))
    assert not tied._one_pass_ok(mk(8, 32), mk(16, 128))            # padding would add more than a quarter
    assert not tied._one_pass_ok(mk(8, 160), mk(64, 128))           # queries longer than passages
    assert not tied._one_pass_ok(mk(8, 32, token_type_ids=torch.zeros(8, 32, dtype=torch.long)), mk(64, 128))
    assert not tied._one_pass_ok(None, mk(64, 128))
    with torch.no_grad():
        assert not tied._one_pass_ok(mk(8, 32), mk(64, 128))
    assert not tied.eval()._one_pass_ok(mk(8, 32), mk(64, 128))
    assert not DRModel(lm_q=lm, lm_p=other, tied=False, **args).train()._one_pass_ok(mk(8, 32), mk(64, 128))


def test_compute_format_resolution(monkeypatch):
    """float16 requests: served as float16 for BERT-family erf-GELU encoders and T5 stacks (inference: round 5; training: round 6),
    as bfloat16 elsewhere (other activations, OM_T5_F16=0, OM_TRAIN_F16=0, a format named without a model)."""
    from types import SimpleNamespace as NS
    from transformers import BertConfig, BertModel, T5Config, T5EncoderModel
    from openmatch_amd import native as N
    from openmatch_amd.encoder import compute_dtype_code, inference_code, training_code, torch_dtype_of
    assert compute_dtype_code(NS(dtype="float16")) == N.OM_F16 and compute_dtype_code(NS(dtype="fp16")) == N.OM_F16
    assert compute_dtype_code(NS(dtype="bfloat16")) == N.OM_BF16 and compute_dtype_code(NS(dtype="float32")) == N.OM_F32
    assert compute_dtype_code(None) == N.OM_F32
    bert = BertModel(BertConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128))
    relu = BertModel(BertConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128, hidden_act="relu"))
    t5 = T5EncoderModel(T5Config(d_model=64, d_ff=128, num_layers=1, num_heads=1, d_kv=64))
    assert inference_code(bert, N.OM_F16, 128) == N.OM_F16 and inference_code(bert, N.OM_F16, 256) == N.OM_F16
    assert inference_code(bert, N.OM_F16, 512) == N.OM_F16
    assert inference_code(relu, N.OM_F16, 128) == N.OM_BF16 and inference_code(t5, N.OM_F16, 128) == N.OM_F16
    monkeypatch.setenv("OM_T5_F16", "0")
    assert inference_code(t5, N.OM_F16, 128) == N.OM_BF16 and inference_code(bert, N.OM_F16, 128) == N.OM_F16
    monkeypatch.delenv("OM_T5_F16")
    assert inference_code(t5, N.OM_BF16, 128) == N.OM_BF16 and inference_code(bert, N.OM_F32, 512) == N.OM_F32
    assert training_code(N.OM_F16) == N.OM_BF16 and training_code(N.OM_F32) == N.OM_F32
    # float16 training is served for BERT-family erf-GELU encoders (round 5) and T5 stacks (round 6), bfloat16 elsewhere
    assert training_code(N.OM_F16, bert) == N.OM_F16 and training_code(N.OM_F16, t5) == N.OM_F16 and training_code(N.OM_F16, relu) == N.OM_BF16
    assert training_code(N.OM_BF16, bert) == N.OM_BF16
    monkeypatch.setenv("OM_T5_F16", "0")
    assert training_code(N.OM_F16, t5) == N.OM_BF16 and training_code(N.OM_F16, bert) == N.OM_F16
    monkeypatch.delenv("OM_T5_F16")
    monkeypatch.setenv("OM_TRAIN_F16", "0")
    assert training_code(N.OM_F16, t5) == N.OM_BF16 and training_code(N.OM_F16, bert) == N.OM_BF16
    monkeypatch.delenv("OM_TRAIN_F16")
    assert torch_dtype_of(N.OM_F16) == torch.float16 and torch_dtype_of(N.OM_BF16) == torch.bfloat16 and torch_dtype_of(N.OM_F32) == torch.float32


def test_packed_weight_cache_survives_deepcopy_and_pickle():
    """openmatch_amd/encoder.py keeps packed device weights (ctypes structs) in the module's __dict__: a deep copy or a
    pickle of a model that has already run must not trip over them -- the copy starts with an empty cache."""
    import copy, ctypes, pickle
    import torch
    from openmatch_amd.encoder import _PACK_CACHE_ATTR, _PackCache
    m = torch.nn.Linear(2, 2)

    class Holder(ctypes.Structure):
        _fields_ = [("p", ctypes.c_void_p)]
    cache = m.__dict__.setdefault(_PACK_CACHE_ATTR, _PackCache())
    cache["k"] = ("v1", Holder())
    m2 = copy.deepcopy(m)
    assert isinstance(m2.__dict__[_PACK_CACHE_ATTR], _PackCache) and len(m2.__dict__[_PACK_CACHE_ATTR]) == 0
    m3 = pickle.loads(pickle.dumps(m))
    assert len(m3.__dict__[_PACK_CACHE_ATTR]) == 0 and torch.equal(m3.weight, m.weight)
    assert len(cache) == 1


def test_deferred_append_positions_are_not_read_before_their_wait():
    """sim_stream_reg_kernel (csrc/search.hip) issues the list-counter atomics of a tile as inline assembly and reads
    their results only behind a hand-written s_waitcnt four units later -- the compiler does not know the destination
    registers are written asynchronously, so a copy it might insert (loop-carried value) would read them too early.
    Compile the translation unit to assembly and check, for every instantiation, that no instruction between an atomic
    and the retire point's wait touches the atomic's destination register."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(REPO, "openmatch_amd", "csrc", "search.hip")
    out = os.path.join(REPO, "build", "search_check.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "-Wno-unused-result", "-w",
                    "--cuda-device-only", "-S", src, "-o", out], check=True)
    text = open(out).read()

    def uses(n, line):
        if any(int(m.group(1)) == n for m in re.finditer(r"\bv(\d+)\b", line)):
            return True
        return any(int(m.group(1)) <= n <= int(m.group(2)) for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line))

    seen = 0
    for nb in (1, 2, 4):
        m = re.search(r"^_Z21sim_stream_reg_kernelIDF16_Li%dE.*?\.amdhsa_kernel" % nb, text, re.S | re.M)
        assert m, nb
        lines = m.group(0).split("\n")
        atomics = [i for i, l in enumerate(lines) if "global_atomic_add" in l]
        assert len(atomics) == nb, (nb, atomics)
        first = next(i for i, l in enumerate(lines) if "s_waitcnt vmcnt(32)" in l)      # a unit's steady-state wait
        header = max(i for i, l in enumerate(lines[:first]) if "Loop Header" in l)
        # the retire point: the only hand-written vmcnt(16) of the kernel
        retire = next(i for i, l in enumerate(lines) if "s_waitcnt vmcnt(16)" in l and i > first)
        for a in atomics:
            reg = int(re.search(r"global_atomic_add v(\d+),", lines[a]).group(1))
            # rest of the loop body behind the atomic (the staging code ends the body), then from the loop header to the wait
            tail_end = next((i for i in range(a + 1, len(lines)) if "s_cbranch" in lines[i] and ("LBB" in lines[i])), a + 40)
            span = list(range(a + 1, tail_end)) + list(range(header, retire))
            early = [lines[i].strip() for i in span if uses(reg, lines[i]) and not lines[i].strip().startswith(";")
                     and "global_atomic_add" not in lines[i]]
            assert not early, (nb, reg, early[:3])
            seen += 1
    assert seen == 7


def test_mfma_loops_keep_their_accumulators_in_agprs():
    """A register-allocation trap of hipcc on gfx950: unless every accumulator tile is pinned with
    asm("" : "+a"(acc)) where control flow joins, one of them may get a VGPR home and be copied in and out of the AGPR file
    around its MFMAs in EVERY loop iteration (32-128 v_accvgpr moves per iteration were found in the generation-6 K loop and
    in the small-batch scan late in round 3).  Compile the translation units of the hot loops to assembly and require that
    no loop holding MFMAs also holds v_accvgpr moves -- except the listed ones, whose moves are their work (the scan's
    filter reads every score) or belong to paths outside the benchmarks (T5 gated-activation training epilogues)."""
    import re
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    units = ["search.hip", "gemm_wide6_bf16.hip", "gemm_tn.hip", "gemm_wide7.hip"]
    # (the allow-list with per-iteration bounds is below)
    os.makedirs(os.path.join(REPO, "build"), exist_ok=True)

    def compile_unit(u):
        out = os.path.join(REPO, "build", "agpr_check_" + u.replace(".hip", ".s"))
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "-Wno-unused-result", "-w",
                        "--cuda-device-only", "-S", os.path.join(REPO, "openmatch_amd", "csrc", u), "-o", out], check=True)
        return open(out).read()

    with ThreadPoolExecutor(max_workers=4) as pool:
        texts = list(pool.map(compile_unit, units))

    def innermost_loops(lines):
        """{header: [mfma count, v_accvgpr count]} of the innermost loops, by LLVM's block annotations (every block of a loop
        names its header; a rotated loop's back edge need not target the header label)."""
        loops, inner, cur = {}, set(), None
        for line in lines:
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", line):
                cur = set()
                m = re.search(r"Header=(BB\d+_\d+)", line)
                if m:
                    cur.add(m.group(1))
                m = re.match(r"^\.L(BB\d+_\d+):.*Loop Header", line)
                if m:
                    cur.add(m.group(1))
                    if "Inner Loop Header" in line:
                        inner.add(m.group(1))
            elif cur is not None:
                m = re.search(r"Parent Loop (BB\d+_\d+)", line)
                if m:
                    cur.add(m.group(1))
                if not line.strip().startswith(";"):
                    for h in cur:
                        c = loops.setdefault(h, [0, 0])
                        c[0] += "v_mfma" in line
                        c[1] += "v_accvgpr" in line
        return {h: c for h, c in loops.items() if h in inner}

    # innermost loops whose accumulator moves are their work, with a bound per iteration
    bounds = {"sim_filter_kernel7": 544,                   # the persistent tile loop holds the filter: every score is read once
              "sim_stream_reg_kernelIDF16_Li1E": 48,       # per TILE (the K steps are unrolled inside): shadow copy + zeroing
              "sim_stream_reg_kernelIDF16_Li2E": 96,
              "sim_stream_reg_kernelIDF16_Li4E": 288,
              "gemm_nt_kernel6IttLi3": 32}                 # gelu_new (T5 v1.1 gated) training epilogues: one tile still bounces
    checked, offenders = 0, []
    for text in texts:
        for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)^\s*\.amdhsa_kernel \1", text, re.S | re.M):
            name = m.group(1)
            for header, (n_mfma, n_move) in innermost_loops(m.group(2).split("\n")).items():
                if not n_mfma:
                    continue
                checked += 1
                limit = next((v for k, v in bounds.items() if k in name), 0)
                if n_move > limit:
                    offenders.append((name[:70], header, n_mfma, n_move, limit))
    assert checked >= 25, checked
    assert not offenders, offenders


def test_token_rows_bound_counts_rows_up_to_last_token():
    """feed.token_rows_bound / encoder.packed_rows_bound: the host-side row bound of om_encoder_forward_packed."""
    import torch
    from openmatch_amd.encoder import packed_rows_bound
    from openmatch_amd.feed import pack_token_batch, token_rows_bound
    L = 128
    lens = [128, 1, 77, 0, 33] * 4
    mask = (torch.arange(L)[None, :] < torch.tensor(lens)[:, None]).long()
    ids = torch.randint(1, 30000, (len(lens), L)) * mask
    want = (sum(n if n else L for n in lens) + 255) // 256 * 256
    assert token_rows_bound(pack_token_batch({"input_ids": ids, "attention_mask": mask})) == want
    assert packed_rows_bound(mask) == want
    holes = mask.clone(); holes[0, 5:100] = 0                  # not a prefix mask: travels as mask8, no lengths -> no bound
    assert token_rows_bound(pack_token_batch({"input_ids": ids, "attention_mask": holes})) is None
    assert packed_rows_bound(holes) == want                    # last unmasked token of row 0 is still 127
    assert token_rows_bound({"input_ids": ids, "attention_mask": mask}) is None
    assert packed_rows_bound(mask[:1, :64]) is None            # fewer than 512 rows


def test_model_batch_and_packed_rows_apply_conditions():
    """feed.model_batch hands the compact batch only to models that widen it themselves; encoder.packed_rows_apply states
    the conditions of om_encoder_forward_packed (include/openmatch_hip.h) on the host."""
    import torch
    from types import SimpleNamespace as NS
    from openmatch_amd import native as N
    from openmatch_amd.encoder import packed_rows_apply
    from openmatch_amd.feed import is_packed, model_batch, pack_token_batch
    from openmatch_amd.modeling import DRModel, RRModel
    L = 16
    mask = (torch.arange(L)[None, :] < torch.tensor([16, 3, 9])[:, None]).long()
    compact = pack_token_batch({"input_ids": torch.randint(1, 100, (3, L)) * mask, "attention_mask": mask})
    assert is_packed(compact)
    assert DRModel.accepts_compact_batches and RRModel.accepts_compact_batches
    assert model_batch(compact, "cpu", NS(accepts_compact_batches=True)) is compact
    wide = model_batch(compact, "cpu", object())            # any other model: the reference's three int64 tensors
    assert set(wide) == {"input_ids", "attention_mask"} and wide["input_ids"].dtype == torch.int64
    assert torch.equal(wide["attention_mask"], mask)

    cfg = NS(arch=N.ARCH_BERT, dtype=N.OM_F16, act=N.ACT_GELU_ERF, hidden=768, ffn=3072, n_layers=12)
    B, Lp = 64, 128
    ok = lambda **kw: packed_rows_apply(NS(**{**vars(cfg), **kw.pop("cfg", {})}), kw.pop("B", B), kw.pop("L", Lp), kw.pop("rows", 4096),
                                        kw.pop("want_hidden", False), kw.pop("pooling", "first"), kw.pop("gated", False))
    assert ok()
    assert not ok(want_hidden=True) and not ok(pooling=None)
    assert not ok(cfg={"dtype": N.OM_F32}) and not ok(cfg={"hidden": 128}) and not ok(cfg={"act": N.ACT_RELU})
    assert ok(cfg={"arch": N.ARCH_T5, "act": N.ACT_RELU, "dtype": N.OM_BF16}) and not ok(cfg={"arch": N.ARCH_T5, "act": N.ACT_RELU}, gated=True)
    assert not ok(rows=256) and not ok(rows=4100) and not ok(rows=B * Lp) and ok(rows=B * Lp - 256)
    assert ok(L=512, rows=4096) and not ok(L=1100, rows=4096)       # (round 6: packed rows at every inference length the encoder takes)
    # the library's run-time switches are part of the answer (om_encoder_packed_supported): with the fused path switched off for an
    # A/B run the compact batch goes to the padded entry instead of failing inside the call
    lib = N.lib()
    for opt, off in ((0, 0), (12, 2)):                       # OM_OPT_ENCODER_FUSED_LN = 0; OM_OPT_GEMM_VARIANT = 2
        default = 1 if opt == 0 else 0
        N.check(lib.om_debug_option(opt, off))
        try:
            assert not ok()
        finally:
            N.check(lib.om_debug_option(opt, default))
    assert ok()
    os.environ["OM_ENCODER_PACKED"] = "0"
    try:
        assert not ok()
    finally:
        del os.environ["OM_ENCODER_PACKED"]


def test_host_side_token_counts_split_and_merge():
    """openmatch_amd.encoder.token_rows_of / rows_bound_of (packed rows in training): per-sequence extents of a host-side mask -- last
    unmasked token + 1, L for an all-masked row, holes counted -- their bound in whole 256-row tiles (None below 512 rows), and the
    gradient-cache chunking keeps them per chunk (trainer/dense_trainer.py:split_dense_inputs)."""
    import torch
    from openmatch_amd.encoder import TOKEN_ROWS_KEY, rows_bound_of, token_rows_of
    from openmatch_amd.trainer.dense_trainer import split_dense_inputs
    m = torch.zeros(6, 128, dtype=torch.long)
    m[0, :128] = 1; m[1, :5] = 1; m[2, ::7] = 1; m[4, :64] = 1; m[5, 100] = 1          # row 3: nothing unmasked
    ext = token_rows_of(m)
    assert ext.tolist() == [128, 5, 127, 128, 64, 101] and ext.dtype == torch.int64 and not ext.is_cuda
    assert rows_bound_of(ext) == 768 and rows_bound_of(553) == 768 and rows_bound_of(512) == 512 and rows_bound_of(511) == 512
    assert rows_bound_of(100) is None and rows_bound_of(None) is None and rows_bound_of(ext[1:3]) is None
    assert token_rows_of(m[0]) is None and token_rows_of("x") is None
    chunks = split_dense_inputs({"passage": {"input_ids": m.clone(), "attention_mask": m, TOKEN_ROWS_KEY: ext, "note": 3}}, 4)
    assert [c["passage"][TOKEN_ROWS_KEY].tolist() for c in chunks] == [[128, 5, 127, 128], [64, 101]]
    assert all("note" not in c["passage"] and c["passage"]["attention_mask"].shape[0] == len(c["passage"][TOKEN_ROWS_KEY]) for c in chunks)
