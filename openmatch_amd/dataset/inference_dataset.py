"""Corpus / query text feed of the encode path (reference: dataset/inference_dataset.py:15-180).

Same classes, constructor arguments and record format (`{"text_id", "input_ids"[, "attention_mask",
"token_type_ids"]}`), different plumbing:

* records are read straight from the file with a plain line reader (no `datasets.load_dataset`
  streaming builder, no Arrow cache): `.json`/`.jsonl` one object per line, `.tsv`/`.txt`
  tab-separated with the column names of `DataArguments.{query,doc}_column_names`;
* the rank partition is the reference's (rank r takes rows [r*B, (r+1)*B) of every B*W block,
  inference_dataset.py:99-115) so shard files and `doc_lookup` order are unchanged;
* with `DataLoader(num_workers > 1)` every worker of the reference replays the WHOLE stream (the
  duplication its docs warn about, docs/dr-msmarco-passage.md:229-231).  Here worker w of n takes the
  blocks b with b % n == w; a DataLoader whose batch size equals `batch_size` then returns the
  batches in exactly the single-worker order.
"""
import json
import os

from torch.utils.data import IterableDataset, get_worker_info

from ..utils import fill_template, find_all_markers


def get_idx(obj):
    """`_id`, else `id`, as a string; None when the record has neither (inference_dataset.py:14-17)."""
    for key in ("_id", "id"):
        value = obj.get(key, None)
        if value:
            return str(value)
    return None


def _jsonl_records(path):
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                yield json.loads(line)


def _tsv_records(path, columns):
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\n").rstrip("\r")
            if not line:
                continue
            cells = line.split("\t")
            yield {name: cell for name, cell in zip(columns, cells)}


class InferenceDataset(IterableDataset):
    """Base class; use `InferenceDataset.load(...)` to get the reader for the file's extension."""

    def __init__(self, tokenizer, data_args, is_query=False, final=True, stream=True, batch_size=1,
                 num_processes=1, process_index=0, cache_dir=None):
        super().__init__()
        self.cache_dir = cache_dir
        self.processed_data_path = getattr(data_args, "processed_data_path", None)
        self.data_files = [data_args.query_path] if is_query else [data_args.corpus_path]
        self.tokenizer = tokenizer
        self.max_len = data_args.q_max_len if is_query else data_args.p_max_len
        self.proc_num = getattr(data_args, "dataset_proc_num", 1)
        self.template = data_args.query_template if is_query else data_args.doc_template
        self.all_markers = find_all_markers(self.template)
        self.stream = stream
        self.final = final
        self.batch_size = batch_size
        self.num_processes = num_processes
        self.process_index = process_index
        self.is_query = is_query
        self.dataset = None        # non-stream mode: {text_id: record}

    @classmethod
    def load(cls, tokenizer, data_args, is_query=False, final=True, stream=True, batch_size=1,
             num_processes=1, process_index=0, cache_dir=None):
        path = data_args.query_path if is_query else data_args.corpus_path
        ext = os.path.splitext(path)[1]
        reader = {".json": JsonlDataset, ".jsonl": JsonlDataset, ".tsv": TsvDataset, ".txt": TsvDataset}.get(ext)
        if reader is None:
            raise ValueError("Unsupported dataset file extension {}".format(ext))
        return reader(tokenizer=tokenizer, data_args=data_args, is_query=is_query, final=final,
                      stream=stream, batch_size=batch_size, num_processes=num_processes,
                      process_index=process_index, cache_dir=cache_dir)

    # -- readers provide this ------------------------------------------------------------------
    def records(self):
        raise NotImplementedError

    def _index_by_id(self):
        table = {}
        for obj in self.records():
            table[get_idx(obj)] = obj
        return table

    # -- tokenisation ---------------------------------------------------------------------------
    def process_one(self, example):
        """One record -> model input.  `final=True`: special tokens, padded to max_len, mask and
        type ids (what the bi-encoder eats); `final=False`: bare ids for the pair builder of the
        re-ranker (inference_dataset.py:86-97)."""
        text = fill_template(self.template, example, self.all_markers, allow_not_found=True)
        enc = self.tokenizer(
            text,
            add_special_tokens=self.final,
            padding="max_length" if self.final else False,
            truncation=True,
            max_length=self.max_len,
            return_attention_mask=self.final,
            return_token_type_ids=self.final,
        )
        return {"text_id": get_idx(example), **enc}

    # -- iteration ------------------------------------------------------------------------------
    def __iter__(self):
        info = get_worker_info()
        n_workers, worker = (info.num_workers, info.id) if info is not None else (1, 0)
        block = self.batch_size * self.num_processes
        lo, hi = self.process_index * self.batch_size, (self.process_index + 1) * self.batch_size
        source = self.records() if self.stream or self.dataset is None else iter(self.dataset.values())
        pos = 0                                 # position inside the current block
        block_id = 0
        for rec in source:
            if lo <= pos < hi and block_id % n_workers == worker:
                yield self.process_one(rec)
            pos += 1
            if pos == block:
                pos, block_id = 0, block_id + 1

    def __getitem__(self, index):
        if self.dataset is None:
            self.dataset = self._index_by_id()
        return self.process_one(self.dataset[index])

    def __len__(self):
        n = sum(1 for _ in self.records())
        full, rest = divmod(n, self.batch_size * self.num_processes)
        mine = min(max(rest - self.process_index * self.batch_size, 0), self.batch_size)
        return full * self.batch_size + mine


class JsonlDataset(InferenceDataset):
    def __init__(self, tokenizer, data_args, **kwargs):
        super().__init__(tokenizer, data_args, **kwargs)
        first = next(self.records(), None)
        self.all_columns = list(first.keys()) if first is not None else []
        if not self.stream:
            self.dataset = self._index_by_id()

    def records(self):
        return _jsonl_records(self.data_files[0])


class TsvDataset(InferenceDataset):
    def __init__(self, tokenizer, data_args, is_query=False, **kwargs):
        super().__init__(tokenizer, data_args, is_query=is_query, **kwargs)
        names = data_args.query_column_names if is_query else data_args.doc_column_names
        self.all_columns = names.split(",")
        if not self.stream:
            self.dataset = self._index_by_id()

    def records(self):
        return _tsv_records(self.data_files[0], self.all_columns)
