"""Encode the queries, search the embedding shards, write a TREC run (reference: driver/retrieve.py:17-76)."""
import logging

from ..arguments import InferenceArguments as EncodingArguments
from ..modeling import DRModelForInference
from ..retriever import Retriever
from ..utils import save_as_trec
from ._common import load_config_and_tokenizer, load_text, parse, setup_logging

logger = logging.getLogger(__name__)


def run(retriever_cls):
    model_args, data_args, encoding_args = parse(EncodingArguments)
    setup_logging(logger, encoding_args, model_args)
    config, tokenizer = load_config_and_tokenizer(model_args, use_fast=False)
    model = DRModelForInference.build(model_args=model_args, config=config, cache_dir=model_args.cache_dir)
    queries = load_text(tokenizer, data_args, encoding_args, model_args, is_query=True)
    retriever = retriever_cls.from_embeddings(model, encoding_args)
    result = retriever.retrieve(queries)
    if encoding_args.local_process_index == 0:
        save_as_trec(result, encoding_args.trec_save_path)


def main():
    run(Retriever)


if __name__ == "__main__":
    main()
