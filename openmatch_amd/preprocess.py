"""Training-set construction for the train -> encode -> search -> re-train cycle
(reference: utils.py:14-101 `SimpleTrainPreProcessor`, scripts/msmarco/build_train.py,
scripts/msmarco/build_hn.py).  Pure host text tooling: it turns (query id, positive ids, negative
ids) triples into the pre-tokenised jsonl groups DRTrainDataset streams.

The collection is addressed by ROW NUMBER, as in the reference (`collection[int(p)]`: MS MARCO
passage ids are row numbers); rows are found through a byte-offset table built in one pass, so the
multi-gigabyte tsv is neither parsed into Arrow nor held in memory.
"""
import json
import os
import random
from array import array

from .utils import fill_template


class SimpleTrainPreProcessor:
    columns = ["text_id", "title", "text"]
    title_field, text_field, query_field = "title", "text", "text"

    def __init__(self, query_file, collection_file, tokenizer, doc_max_len=128, query_max_len=32,
                 doc_template=None, query_template=None, allow_not_found=False):
        self.query_file, self.collection_file, self.tokenizer = query_file, collection_file, tokenizer
        self.doc_max_len, self.query_max_len = doc_max_len, query_max_len
        self.doc_template, self.query_template = doc_template, query_template
        self.allow_not_found = allow_not_found
        self.queries = self.read_queries(query_file)
        self.offsets = array("q")
        with open(collection_file, "rb") as f:
            pos = 0
            for line in f:
                self.offsets.append(pos)
                pos += len(line)
        self._fh = None

    @staticmethod
    def read_queries(path):
        table = {}
        with open(path, encoding="utf-8") as f:
            for line in f:
                qid, text = line.rstrip("\n").split("\t")[:2]
                table[qid] = text
        return table

    @staticmethod
    def read_qrel(path):
        """{query id: [relevant doc ids]} from a TREC qrels tsv (`qid 0 docid 1`)."""
        qrel = {}
        with open(path, encoding="utf-8") as f:
            for line in f:
                parts = line.split()
                if len(parts) != 4:
                    continue
                qid, _, docid, rel = parts
                assert rel == "1", "binary judgements expected"
                qrel.setdefault(qid, []).append(docid)
        return qrel

    def _row(self, index):
        if self._fh is None:                       # one handle per process (multiprocessing-safe)
            self._fh = open(self.collection_file, "rb")
        self._fh.seek(self.offsets[index])
        cells = self._fh.readline().decode("utf-8").rstrip("\n").rstrip("\r").split("\t")
        return {name: (cells[i] if i < len(cells) else None) for i, name in enumerate(self.columns)}

    def _ids(self, text, max_len):
        return self.tokenizer.encode(text, add_special_tokens=False, max_length=max_len, truncation=True)

    def get_query(self, q):
        text = self.queries[q]
        if self.query_template is not None:
            text = fill_template(self.query_template, data={self.query_field: text}, allow_not_found=self.allow_not_found)
        return self._ids(text, self.query_max_len)

    def get_passage(self, p):
        entry = self._row(int(p))
        if self.doc_template is None:
            content = (entry[self.title_field] or "") + self.tokenizer.sep_token + entry[self.text_field]
        else:
            content = fill_template(self.doc_template, data=entry, allow_not_found=self.allow_not_found)
        return self._ids(content, self.doc_max_len)

    def process_one(self, train):
        q, positives, negatives = train
        return json.dumps({"query": self.get_query(q), "positives": [self.get_passage(p) for p in positives],
                           "negatives": [self.get_passage(n) for n in negatives]})

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_fh"] = None
        return state


class SimpleCollectionPreProcessor:
    """One collection line `id <sep> field <sep> field ...` -> `{"text_id", "text": [token ids]}` as a json string,
    the fields joined by the tokenizer's separator token and cut to max_length (reference utils.py:104-124)."""

    def __init__(self, tokenizer, separator="\t", max_length=128):
        self.tokenizer, self.separator, self.max_length = tokenizer, separator, max_length

    def process_line(self, line):
        cells = line.strip().split(self.separator)
        ids = self.tokenizer.encode(self.tokenizer.sep_token.join(cells[1:]), add_special_tokens=False,
                                    max_length=self.max_length, truncation=True)
        return json.dumps({"text_id": cells[0], "text": ids})


def negatives_from_run(rank_file, relevance, n_sample, depth, rng=random):
    """(qid, positives, sampled hard negatives) per query of a TREC run file: the first `depth`
    non-relevant hits, shuffled, `n_sample` of them kept (build_hn.py:13-38)."""
    current, pool = None, []

    def flush():
        kept = pool[:depth]
        rng.shuffle(kept)
        return current, relevance[current], kept[:n_sample]

    with open(rank_file) as f:
        for line in f:
            parts = line.split()
            if len(parts) < 3:
                continue
            q, p = parts[0], parts[2]
            if q != current:
                if current is not None:
                    yield flush()
                current, pool = q, []
            if p not in relevance[q]:
                pool.append(p)
    if current is not None:
        yield flush()


def negatives_from_list(negative_file, relevance, n_sample, rng=random):
    """(qid, positives, sampled negatives) from `qid \\t neg,neg,...` lines (build_train.py:30-34)."""
    with open(negative_file) as f:
        for line in f:
            q, negs = line.rstrip("\n").split("\t")
            negs = negs.split(",")
            rng.shuffle(negs)
            yield q, relevance[q], negs[:n_sample]


def write_shards(lines, save_to, shard_size, suffix):
    """`split{NN}{suffix}` files of at most shard_size lines; returns the paths."""
    os.makedirs(save_to, exist_ok=True)
    paths, f, count = [], None, 0
    for line in lines:
        if f is None:
            paths.append(os.path.join(save_to, "split%02d%s" % (len(paths), suffix)))
            f = open(paths[-1], "w")
        f.write(line + "\n")
        count += 1
        if count == shard_size:
            f.close()
            f, count = None, 0
    if f is not None:
        f.close()
    return paths
