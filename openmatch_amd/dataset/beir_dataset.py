"""BEIR-format evaluation sets (reference: dataset/beir_dataset.py:10-97): a directory with
`corpus.jsonl` ({"_id", "title", "text"}), `queries.jsonl` ({"_id", "text"}) and `qrels/test.tsv`
(header line, then `query-id  corpus-id  score`).

The reference's classes cannot run as published (the corpus reader asks DataArguments for a
`template` field it does not have; the driver evaluates the return value of a method that
returns None), so this is the behaviour they describe rather than a port: per-example records
like InferenceDataset's (`text_id` + padded token ids), empty titles replaced by "-", only the
queries that have relevance judgements, and the encode path's rank partition.
"""
import json
import os

from torch.utils.data import IterableDataset, get_worker_info


def _jsonl(path):
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                yield json.loads(line)


def load_qrels(path):
    """{qid: {docid: int relevance}} from a BEIR qrels tsv (first line is a header)."""
    qrel = {}
    with open(path, "r", encoding="utf-8") as f:
        next(f, None)
        for line in f:
            parts = line.split()
            if len(parts) < 3:
                continue
            qid, docid, rel = parts[0], parts[1], int(parts[2])
            qrel.setdefault(qid, {})[docid] = rel
    return qrel


class _BEIRText(IterableDataset):
    def __init__(self, tokenizer, file_path, max_len, batch_size=1, num_processes=1, process_index=0):
        super().__init__()
        self.tokenizer = tokenizer
        self.file_path = file_path
        self.max_len = max_len
        self.batch_size, self.num_processes, self.process_index = batch_size, num_processes, process_index

    def records(self):
        return _jsonl(self.file_path)

    def text_of(self, example):
        raise NotImplementedError

    def process_one(self, example):
        enc = self.tokenizer(self.text_of(example), padding="max_length", truncation=True, max_length=self.max_len)
        return {"text_id": str(example["_id"]), **enc}

    def __iter__(self):
        info = get_worker_info()
        n_workers, worker = (info.num_workers, info.id) if info is not None else (1, 0)
        block = self.batch_size * self.num_processes
        lo, hi = self.process_index * self.batch_size, (self.process_index + 1) * self.batch_size
        pos = block_id = 0
        for rec in self.records():
            if lo <= pos < hi and block_id % n_workers == worker:
                yield self.process_one(rec)
            pos += 1
            if pos == block:
                pos, block_id = 0, block_id + 1


class BEIRQueryDataset(_BEIRText):
    def __init__(self, tokenizer, data_args, file_path, qids, cache_dir=None, **partition):
        super().__init__(tokenizer, file_path, data_args.q_max_len, **partition)
        self.qids = set(qids)
        self.template = getattr(data_args, "query_template", "<text>") or "<text>"

    def records(self):
        return (r for r in _jsonl(self.file_path) if str(r["_id"]) in self.qids)

    def text_of(self, example):
        return self.template.replace("<text>", example["text"])


class BEIRCorpusDataset(_BEIRText):
    def __init__(self, tokenizer, data_args, file_path, cache_dir=None, **partition):
        super().__init__(tokenizer, file_path, data_args.p_max_len, **partition)
        template = getattr(data_args, "template", None) or data_args.doc_template
        self.template = getattr(tokenizer, template, template)     # e.g. "sep_token"-style indirection

    def text_of(self, example):
        title = (example.get("title") or "").strip() or "-"
        return self.template.replace("<title>", title).replace("<text>", example["text"])


class BEIRDataset:
    def __init__(self, tokenizer, data_args, cache_dir=None, **partition):
        root = data_args.data_dir
        self.corpus_dataset = BEIRCorpusDataset(tokenizer, data_args, os.path.join(root, "corpus.jsonl"), cache_dir, **partition)
        self.qrel = load_qrels(os.path.join(root, "qrels", "test.tsv"))
        self.query_dataset = BEIRQueryDataset(tokenizer, data_args, os.path.join(root, "queries.jsonl"),
                                              list(self.qrel.keys()), cache_dir, **partition)
