"""Training text feed (reference: dataset/train_dataset.py:16-195).

Input files are the reference's pre-tokenised jsonl (`{"query": [ids], "positives": [[ids]..],
"negatives": [[ids]..]}`, one training group per line; `train_path` or every `*.jsonl` of
`train_dir`).  The example-building rules are the reference's (which positive, which window of
negatives per epoch: train_dataset.py:72-107, 158-176); the stream itself is a plain line reader
with a seeded shuffle buffer instead of `datasets` streaming, and DataLoader workers split the
groups between them instead of replaying the stream.
"""
import glob
import json
import os
import random

from torch.utils.data import IterableDataset, get_worker_info

SHUFFLE_BUFFER = 10_000      # the reference shuffles its stream through a 10 000-example buffer


def wrap_ids(tokenizer, ids, max_length):
    """Special tokens around a list of token ids, cut to `max_length` -- what the reference gets from
    `tokenizer.encode_plus(ids, truncation=..., max_length=..., padding=False)` under transformers 4.x
    (train_dataset.py:60-67).  transformers 5 removed that entry point for id lists, so the rule is
    applied directly: [CLS] ids [SEP] for BERT-like vocabularies, ids </s> for T5-like ones."""
    if hasattr(tokenizer, "prepare_for_model"):          # transformers 4.x: the call the reference makes
        return dict(tokenizer.prepare_for_model(ids, truncation="only_first", max_length=max_length,
                                                padding=False, return_attention_mask=False,
                                                return_token_type_ids=False))
    cls_id, sep_id = getattr(tokenizer, "cls_token_id", None), getattr(tokenizer, "sep_token_id", None)
    if cls_id is not None and sep_id is not None:
        return {"input_ids": [cls_id] + list(ids)[:max_length - 2] + [sep_id]}
    eos_id = getattr(tokenizer, "eos_token_id", None)
    if eos_id is not None:
        return {"input_ids": list(ids)[:max_length - 1] + [eos_id]}
    return {"input_ids": list(ids)[:max_length]}


def _group_stream(files):
    for path in files:
        with open(path, "r", encoding="utf-8") as f:
            for line in f:
                line = line.strip()
                if line:
                    yield json.loads(line)


def _buffer_shuffle(stream, seed, size=SHUFFLE_BUFFER):
    """Reservoir-style streaming shuffle: fill a buffer, then emit a random slot per new item."""
    rng = random.Random(seed)
    buf = []
    for item in stream:
        if len(buf) < size:
            buf.append(item)
            continue
        j = rng.randrange(size)
        yield buf[j]
        buf[j] = item
    rng.shuffle(buf)
    yield from buf


class TrainDataset(IterableDataset):
    def __init__(self, tokenizer, data_args, trainer=None, shuffle_seed=None, cache_dir=None):
        super().__init__()
        self.tokenizer = tokenizer
        self.data_args = data_args
        self.q_max_len = data_args.q_max_len
        self.p_max_len = data_args.p_max_len
        self.proc_num = getattr(data_args, "dataset_proc_num", 1)
        self.trainer = trainer
        self.shuffle_seed = shuffle_seed
        self.cache_dir = cache_dir
        self.data_files = self._data_files(data_args)

    def _data_files(self, data_args):
        if getattr(data_args, "train_dir", None) is None:
            return [data_args.train_path]
        return sorted(glob.glob(os.path.join(data_args.train_dir, "*.jsonl")))

    def __len__(self):
        count = 0
        for path in self.data_files:
            with open(path, "rb") as f:
                count += sum(1 for _ in f)
        return count

    # -- epoch bookkeeping ------------------------------------------------------------------------
    def _epoch_and_seed(self):
        """(epoch, hashed seed) read from the attached trainer exactly when iteration starts
        (train_dataset.py:115-119); evaluation datasets have neither."""
        if self.trainer is None:
            return 0, None
        return int(self.trainer.state.epoch), hash(self.trainer.args.seed)

    def groups(self, epoch):
        stream = _group_stream(self.data_files)
        if self.shuffle_seed is not None:
            stream = _buffer_shuffle(stream, self.shuffle_seed + epoch)
        info = get_worker_info()
        if info is not None and info.num_workers > 1:
            stream = (g for i, g in enumerate(stream) if i % info.num_workers == info.id)
        return stream

    def build(self, group, epoch, hashed_seed):
        raise NotImplementedError

    def __iter__(self):
        epoch, hashed_seed = self._epoch_and_seed()
        for group in self.groups(epoch):
            yield self.build(group, epoch, hashed_seed)


def _pick_positive(positives, epoch, hashed_seed, fixed):
    if fixed or hashed_seed is None:
        return positives[0]
    return positives[(hashed_seed + epoch) % len(positives)]


class DRTrainDataset(TrainDataset):
    """One query + `train_n_passages` passages (first the positive) per example; QPCollator pads."""

    def __init__(self, tokenizer, data_args, trainer=None, shuffle_seed=None, cache_dir=None):
        super().__init__(tokenizer, data_args, trainer, shuffle_seed, cache_dir)
        self.neg_num = data_args.train_n_passages - 1

    def create_one_example(self, ids, is_query=False):
        return wrap_ids(self.tokenizer, ids, self.q_max_len if is_query else self.p_max_len)

    def _pick_negatives(self, negatives, epoch, hashed_seed):
        want = self.neg_num
        if len(negatives) < want:                       # too few: sample with replacement / repeat
            if hashed_seed is not None:
                return random.choices(negatives, k=want)
            return (list(negatives) * 2)[:want]
        if self.data_args.train_n_passages == 1:
            return []
        if self.data_args.negative_passage_no_shuffle:
            return negatives[:want]
        # a window that advances by `want` every epoch over a (seeded) permutation of the negatives
        offset = epoch * want % len(negatives)
        pool = list(negatives)
        if hashed_seed is not None:
            random.Random(hashed_seed).shuffle(pool)
        return (pool * 2)[offset:offset + want]

    def build(self, group, epoch, hashed_seed):
        query = self.create_one_example(group["query"], is_query=True)
        positive = _pick_positive(group["positives"], epoch, hashed_seed, self.data_args.positive_passage_no_shuffle)
        passages = [self.create_one_example(positive)]
        passages += [self.create_one_example(n) for n in self._pick_negatives(group["negatives"], epoch, hashed_seed)]
        assert len(passages) == self.data_args.train_n_passages
        return {"query": query, "passages": passages}


class DREvalDataset(DRTrainDataset):
    def __init__(self, tokenizer, data_args, cache_dir=None):
        super().__init__(tokenizer, data_args, None, cache_dir=cache_dir)

    def _data_files(self, data_args):
        return [data_args.eval_path]


class RRTrainDataset(TrainDataset):
    """One (query, positive) pair and one (query, negative) pair per example; PairCollator pads."""

    def __init__(self, tokenizer, data_args, trainer=None, shuffle_seed=None, cache_dir=None):
        super().__init__(tokenizer, data_args, trainer, shuffle_seed, cache_dir)
        self.neg_num = 1

    def create_one_example(self, qry_ids, psg_ids):
        # one concatenated sequence, as in the reference (train_dataset.py:139-147)
        return wrap_ids(self.tokenizer, list(qry_ids) + list(psg_ids), self.q_max_len + self.p_max_len + 2)

    def build(self, group, epoch, hashed_seed):
        query = group["query"]
        positive = _pick_positive(group["positives"], epoch, hashed_seed, self.data_args.positive_passage_no_shuffle)
        negatives = group["negatives"]
        negative = negatives[0] if hashed_seed is None else negatives[(hashed_seed + epoch) % len(negatives)]
        # the raw `query` column rides along, as it does through the reference's `datasets.map`
        return {"query": query, "pos_pair": self.create_one_example(query, positive),
                "neg_pair": self.create_one_example(query, negative)}


class RREvalDataset(RRTrainDataset):
    def __init__(self, tokenizer, data_args, cache_dir=None):
        super().__init__(tokenizer, data_args, None, cache_dir=cache_dir)

    def _data_files(self, data_args):
        return [data_args.eval_path]
