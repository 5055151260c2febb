"""Encode a corpus into per-rank embedding shards (reference: driver/build_index.py:13-51).

    python -m openmatch.driver.build_index --model_name_or_path CKPT --corpus_path corpus.tsv \\
        --doc_template "<title> <text>" --output_dir EMB --per_device_eval_batch_size 256 --p_max_len 128
"""
from ..arguments import InferenceArguments as EncodingArguments
from ..modeling import DRModelForInference
from ..retriever import Retriever
from ._common import load_config_and_tokenizer, load_text, parse


def main():
    model_args, data_args, encoding_args = parse(EncodingArguments)
    config, tokenizer = load_config_and_tokenizer(model_args)
    model = DRModelForInference.build(model_args=model_args, config=config, cache_dir=model_args.cache_dir)
    corpus = load_text(tokenizer, data_args, encoding_args, model_args, is_query=False)
    Retriever.build_embeddings(model, corpus, encoding_args)


if __name__ == "__main__":
    main()
