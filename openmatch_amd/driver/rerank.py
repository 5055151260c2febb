"""Re-score the top of a TREC run with a cross-encoder (reference: driver/rerank.py:17-87)."""
import logging

from ..arguments import InferenceArguments
from ..dataset import InferenceDataset
from ..modeling import RRModel
from ..retriever import Reranker
from ..utils import load_from_trec, save_as_trec
from ._common import load_config_and_tokenizer, parse, setup_logging

logger = logging.getLogger(__name__)


def main():
    model_args, data_args, inference_args = parse(InferenceArguments)
    setup_logging(logger, inference_args, model_args)
    config, tokenizer = load_config_and_tokenizer(model_args, use_fast=False)
    model = RRModel.build(model_args=model_args, tokenizer=tokenizer, config=config, cache_dir=model_args.cache_dir)
    load = lambda is_query: InferenceDataset.load(tokenizer=tokenizer, data_args=data_args, final=False,
                                                  is_query=is_query, stream=False, cache_dir=model_args.cache_dir)
    queries, corpus = load(True), load(False)
    run = load_from_trec(inference_args.trec_run_path, max_len_per_q=inference_args.reranking_depth)
    result = Reranker(model, tokenizer, corpus, inference_args).rerank(queries, run)
    if inference_args.local_process_index == 0:
        save_as_trec(result, inference_args.trec_save_path)


if __name__ == "__main__":
    main()
