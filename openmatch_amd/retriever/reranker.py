"""`Reranker` with the reference's surface (src/openmatch/retriever/reranker.py:63-133): scores
every (query, doc) pair of a first-stage run with the cross-encoder and returns a run dict.
Pairs are independent, so ranks are plain replicas (no collective on the data path); per-rank
TREC files are merged on rank 0 exactly as the reference does."""
import logging
import os
from contextlib import nullcontext
from typing import Dict

import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, IterableDataset
from tqdm import tqdm
from transformers.trainer_pt_utils import IterableDatasetShard

from ..dataset.data_collator import RRInferenceCollator
from ..utils import load_from_trec, merge_retrieval_results_by_score, save_as_trec
from ..feed import model_batch

logger = logging.getLogger(__name__)


def encode_pair(tokenizer, item1, item2, max_len_1=32, max_len_2=128):
    """Query and document token-id lists -> ONE sequence, as the reference builds it
    (reranker.py:23-29: `tokenizer.encode_plus(item1 + item2, truncation='longest_first',
    padding='max_length', max_length=max_len_1 + max_len_2 + 2)`): the two id lists are CONCATENATED, so
    the model sees [CLS] q d [SEP] (no separator between them), token_type_ids all 0, truncated from
    the right, padded to 162 -- the same format `RRTrainDataset.create_one_example` trains on
    (train_dataset.py:139-147).  transformers 5 has no id-list entry point any more, so the 4.x rule is
    applied directly when `prepare_for_model` is missing."""
    n = max_len_1 + max_len_2 + 2
    ids = list(item1) + list(item2)
    if hasattr(tokenizer, "prepare_for_model"):
        return tokenizer.prepare_for_model(ids, truncation="longest_first", padding="max_length", max_length=n)
    cls_id, sep_id = getattr(tokenizer, "cls_token_id", None), getattr(tokenizer, "sep_token_id", None)
    if cls_id is not None and sep_id is not None:
        ids = [cls_id] + ids[:n - 2] + [sep_id]
    else:                                      # T5-like vocabulary: ids </s>
        ids = ids[:n - 1] + [tokenizer.eos_token_id]
    pad = n - len(ids)
    pad_id = tokenizer.pad_token_id if tokenizer.pad_token_id is not None else 0
    out = {"input_ids": ids + [pad_id] * pad, "attention_mask": [1] * len(ids) + [0] * pad}
    if cls_id is not None and sep_id is not None:
        out["token_type_ids"] = [0] * n
    return out


def add_to_result_dict(result_dicts, qids, dids, scores):
    for qid, did, score in zip(qids, dids, scores):
        result_dicts.setdefault(qid, {})[did] = float(score)


class RRPredictDataset(IterableDataset):
    def __init__(self, tokenizer, query_dataset, corpus_dataset, run: Dict[str, Dict[str, float]]):
        super().__init__()
        self.tokenizer, self.query_dataset, self.corpus_dataset, self.run = tokenizer, query_dataset, corpus_dataset, run

    def __iter__(self):
        for qid, hits in self.run.items():
            for did in hits:
                yield {"query_id": qid, "doc_id": did,
                       **encode_pair(self.tokenizer, self.query_dataset[qid]["input_ids"],
                                     self.corpus_dataset[did]["input_ids"], self.query_dataset.max_len,
                                     self.corpus_dataset.max_len)}


class Reranker:
    def __init__(self, model, tokenizer, corpus_dataset, args):
        logger.info("Initializing reranker")
        self.tokenizer, self.corpus_dataset, self.args = tokenizer, corpus_dataset, args
        self.model = model.to(self.args.device)
        self.model.eval()

    def rerank(self, query_dataset, run: Dict[str, Dict[str, float]], pair_dataset=None):
        """`pair_dataset` (optional) replaces the tokenising RRPredictDataset with any iterable of
        {"query_id", "doc_id", input_ids, attention_mask, token_type_ids} items."""
        a = self.args
        result: Dict[str, Dict[str, float]] = {}
        dataset = pair_dataset if pair_dataset is not None else RRPredictDataset(
            self.tokenizer, query_dataset, self.corpus_dataset, run)
        if a.world_size > 1 and isinstance(dataset, IterableDataset):
            dataset = IterableDatasetShard(dataset, batch_size=a.per_device_eval_batch_size, drop_last=False,
                                           num_processes=a.world_size, process_index=a.process_index)
        loader = DataLoader(dataset, batch_size=a.eval_batch_size, collate_fn=RRInferenceCollator(),
                            num_workers=a.dataloader_num_workers, pin_memory=a.dataloader_pin_memory)
        cast = torch.autocast("cuda", dtype=torch.float16) if getattr(a, "fp16", False) else nullcontext()
        with torch.no_grad():
            for qids, dids, batch in tqdm(loader, desc="Reranking", disable=a.local_process_index > 0):
                with cast:
                    batch = model_batch(batch, a.device, self.model)
                    out = self.model.encode(batch)
                if out.dim() == 2 and out.shape[1] == 2:
                    out = F.log_softmax(out, dim=1)[:, 1]
                add_to_result_dict(result, qids, dids, out.reshape(-1).float().cpu().numpy())
        if a.world_size > 1:
            save_as_trec(result, a.trec_save_path + ".rank.{}".format(a.process_index))
            torch.distributed.barrier()
            if a.process_index == 0:
                parts = [load_from_trec(a.trec_save_path + ".rank.{}".format(i)) for i in range(a.world_size)]
                result = merge_retrieval_results_by_score(parts)
                for i in range(a.world_size):
                    os.remove(a.trec_save_path + ".rank.{}".format(i))
            torch.distributed.barrier()
        return result
