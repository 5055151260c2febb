"""Encode a BEIR-format corpus, search it with its test queries, print nDCG@10
(reference: driver/retrieve_beir.py:14-85; `--data_dir` holds corpus.jsonl, queries.jsonl, qrels/test.tsv)."""
import logging

from ..arguments import InferenceArguments as EncodingArguments
from ..dataset import BEIRDataset
from ..modeling import DRModelForInference
from ..retriever import Retriever
from ..utils import eval_ndcg, save_as_trec
from ._common import load_config_and_tokenizer, parse, setup_logging

logger = logging.getLogger(__name__)


def main():
    model_args, data_args, encoding_args = parse(EncodingArguments)
    setup_logging(logger, encoding_args, model_args)
    config, tokenizer = load_config_and_tokenizer(model_args, use_fast=False)
    model = DRModelForInference.build(model_args=model_args, config=config, cache_dir=model_args.cache_dir)
    beir = BEIRDataset(tokenizer=tokenizer, data_args=data_args, cache_dir=model_args.cache_dir,
                       batch_size=encoding_args.per_device_eval_batch_size,
                       num_processes=encoding_args.world_size, process_index=encoding_args.process_index)
    retriever = Retriever.build_all(model, beir.corpus_dataset, encoding_args)
    run = retriever.retrieve(beir.query_dataset)
    if encoding_args.local_process_index == 0:
        if encoding_args.trec_save_path:
            save_as_trec(run, encoding_args.trec_save_path)
        scores = eval_ndcg(beir.qrel, run, cutoff=10)
        print("{:25s}{:8s}{:.4f}".format("ndcg_cut_10", "all", scores["all"]))
    return run


if __name__ == "__main__":
    main()
