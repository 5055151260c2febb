"""Batch collators with the reference's names and outputs (src/openmatch/dataset/data_collator.py)."""
from dataclasses import dataclass

from transformers import DataCollatorWithPadding, DefaultDataCollator
from ..feed import pack_token_batch


def _flatten(groups):
    return [x for g in groups for x in g] if groups and isinstance(groups[0], list) else groups


@dataclass
class QPCollator(DataCollatorWithPadding):
    """List of {"query": enc, "passages": [enc, ...]} -> (queries, passages), each padded to its
    fixed maximum length, so every batch has the static [B,32] / [B*n,128] shapes the HIP encoder
    is tuned for (reference :8-40)."""
    max_q_len: int = 32
    max_p_len: int = 128

    def __call__(self, features):
        queries = _flatten([f["query"] for f in features])
        passages = _flatten([f["passages"] for f in features])
        pad = lambda items, n: self.tokenizer.pad(items, padding="max_length", max_length=n, return_tensors="pt")
        return pad(queries, self.max_q_len), pad(passages, self.max_p_len)


@dataclass
class PairCollator(DataCollatorWithPadding):
    """List of {"pos_pair": enc, "neg_pair": enc} -> (positive pairs, negative pairs), both padded to
    q_max_len + p_max_len + 2 (the cross-encoder's fixed [B,162] shape; reference :43-75)."""
    max_q_len: int = 32
    max_p_len: int = 128

    def __call__(self, features):
        n = self.max_q_len + self.max_p_len + 2
        pad = lambda items: self.tokenizer.pad(items, padding="max_length", max_length=n, return_tensors="pt")
        return pad(_flatten([f["pos_pair"] for f in features])), pad(_flatten([f["neg_pair"] for f in features]))


@dataclass
class DRInferenceCollator(DefaultDataCollator):
    """(text ids, tensor batch) for the encoding loops (reference :78-83)."""

    def __call__(self, features):
        text_ids = [f["text_id"] for f in features]
        return text_ids, pack_token_batch(super().__call__(features))      # compact wire format (feed.py)


@dataclass
class RRInferenceCollator(DefaultDataCollator):
    """(query ids, doc ids, tensor batch) for the re-ranking loop (reference :86-91)."""

    def __call__(self, features):
        query_ids = [f["query_id"] for f in features]
        doc_ids = [f["doc_id"] for f in features]
        keep = [{k: v for k, v in f.items() if k not in ("query_id", "doc_id")} for f in features]
        return query_ids, doc_ids, pack_token_batch(super().__call__(keep))
