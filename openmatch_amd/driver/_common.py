"""Shared start-up of the drivers: argument parsing (a lone *.json path or ordinary flags, as in
the reference drivers), logging, config + tokenizer loading."""
import logging
import os
import sys

from transformers import AutoConfig, AutoTokenizer, HfArgumentParser

from ..arguments import DataArguments, ModelArguments


def parse(run_args_cls):
    parser = HfArgumentParser((ModelArguments, DataArguments, run_args_cls))
    if len(sys.argv) == 2 and sys.argv[1].endswith(".json"):
        return parser.parse_json_file(json_file=os.path.abspath(sys.argv[1]))
    return parser.parse_args_into_dataclasses()


def setup_logging(logger, run_args, model_args, what="Encoding"):
    logging.basicConfig(
        format="%(asctime)s - %(levelname)s - %(name)s -   %(message)s",
        datefmt="%m/%d/%Y %H:%M:%S",
        level=logging.INFO if run_args.local_rank in [-1, 0] else logging.WARN,
    )
    logger.warning(
        "Process rank: %s, device: %s, n_gpu: %s, distributed training: %s, 16-bits training: %s",
        run_args.local_rank, run_args.device, run_args.n_gpu, bool(run_args.local_rank != -1), run_args.fp16)
    logger.info("%s parameters %s", what, run_args)
    logger.info("MODEL parameters %s", model_args)


def load_config_and_tokenizer(model_args, use_fast=None):
    config = AutoConfig.from_pretrained(
        model_args.config_name if model_args.config_name else model_args.model_name_or_path,
        num_labels=1, cache_dir=model_args.cache_dir)
    kwargs = {} if use_fast is None else {"use_fast": use_fast}
    tokenizer = AutoTokenizer.from_pretrained(
        model_args.tokenizer_name if model_args.tokenizer_name else model_args.model_name_or_path,
        cache_dir=model_args.cache_dir, **kwargs)
    return config, tokenizer


def load_text(tokenizer, data_args, run_args, model_args, is_query):
    from ..dataset import InferenceDataset
    return InferenceDataset.load(
        tokenizer=tokenizer, data_args=data_args, is_query=is_query, stream=True,
        batch_size=run_args.per_device_eval_batch_size, num_processes=run_args.world_size,
        process_index=run_args.process_index, cache_dir=model_args.cache_dir)
