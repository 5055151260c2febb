"""Pairwise cross-encoder training (reference: driver/train_rr.py:19-102)."""
import logging
import os

from transformers import set_seed

from ..arguments import RRTrainingArguments as TrainingArguments
from ..dataset import PairCollator, RREvalDataset, RRTrainDataset
from ..modeling import RRModel
from ..trainer import RRTrainer
from ._common import load_config_and_tokenizer, parse, setup_logging

logger = logging.getLogger(__name__)


def main():
    model_args, data_args, training_args = parse(TrainingArguments)
    out = training_args.output_dir
    if os.path.exists(out) and os.listdir(out) and training_args.do_train and not training_args.overwrite_output_dir:
        raise ValueError(
            f"Output directory ({out}) already exists and is not empty. Use --overwrite_output_dir to overcome.")
    setup_logging(logger, training_args, model_args, what="Training/evaluation")
    set_seed(training_args.seed)
    config, tokenizer = load_config_and_tokenizer(model_args, use_fast=False)
    model = RRModel.build(model_args, data_args, training_args, tokenizer=tokenizer, config=config,
                          cache_dir=model_args.cache_dir)
    cache = data_args.data_cache_dir or model_args.cache_dir
    train_dataset = RRTrainDataset(tokenizer, data_args, shuffle_seed=training_args.seed, cache_dir=cache)
    eval_dataset = RREvalDataset(tokenizer, data_args, cache_dir=cache) if data_args.eval_path is not None else None
    trainer = RRTrainer(
        model=model, args=training_args, tokenizer=tokenizer, train_dataset=train_dataset, eval_dataset=eval_dataset,
        data_collator=PairCollator(tokenizer, max_p_len=data_args.p_max_len, max_q_len=data_args.q_max_len))
    train_dataset.trainer = trainer
    trainer.train()
    trainer.save_model()
    if trainer.is_world_process_zero():
        tokenizer.save_pretrained(training_args.output_dir)


if __name__ == "__main__":
    main()
