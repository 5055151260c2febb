"""`qid \\t neg,neg,...` negative lists (e.g. BM25 negatives) -> training groups (same flags as the
reference's scripts/msmarco/build_train.py)."""
import random
from argparse import ArgumentParser
from multiprocessing import Pool

from transformers import AutoTokenizer

from openmatch.preprocess import SimpleTrainPreProcessor, negatives_from_list, write_shards


def main():
    ap = ArgumentParser()
    for name in ("tokenizer_name", "negative_file", "qrels", "queries", "collection", "save_to"):
        ap.add_argument("--" + name, required=True)
    ap.add_argument("--doc_template", type=str, default=None)
    ap.add_argument("--query_template", type=str, default=None)
    ap.add_argument("--truncate", type=int, default=128)
    ap.add_argument("--n_sample", type=int, default=30)
    ap.add_argument("--mp_chunk_size", type=int, default=500)
    ap.add_argument("--shard_size", type=int, default=45000)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--workers", type=int, default=None)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    qrel = SimpleTrainPreProcessor.read_qrel(args.qrels)
    tokenizer = AutoTokenizer.from_pretrained(args.tokenizer_name, use_fast=True)
    proc = SimpleTrainPreProcessor(args.queries, args.collection, tokenizer, doc_max_len=args.truncate,
                                   doc_template=args.doc_template, query_template=args.query_template,
                                   allow_not_found=True)
    triples = negatives_from_list(args.negative_file, qrel, args.n_sample, rng)
    if args.workers == 0:
        write_shards(map(proc.process_one, triples), args.save_to, args.shard_size, ".jsonl")
    else:
        with Pool(args.workers) as pool:
            write_shards(pool.imap(proc.process_one, triples, chunksize=args.mp_chunk_size), args.save_to,
                         args.shard_size, ".jsonl")


if __name__ == "__main__":
    main()
