"""Flag dataclasses of the toolkit.  Field names, types and defaults are the reference's
(src/openmatch/arguments.py:8-191) so existing command lines and `*.json` argument files keep
working under `HfArgumentParser`; only the help strings are ours."""
from dataclasses import dataclass, field
from typing import List, Optional

from transformers import TrainingArguments


def _f(default, help=None):
    return field(default=default, metadata={"help": help} if help else {})


@dataclass
class ModelArguments:
    model_name_or_path: str = field(metadata={"help": "checkpoint directory or hub id of the encoder"})
    target_model_path: str = _f(None, "target model path for the re-ranker")
    config_name: Optional[str] = _f(None, "config to load when it differs from the model")
    tokenizer_name: Optional[str] = _f(None, "tokenizer to load when it differs from the model")
    cache_dir: Optional[str] = _f(None, "download cache for pretrained files")
    # bi-encoder structure
    untie_encoder: bool = _f(False, "separate query and passage encoders")
    feature: str = _f("last_hidden_state", "which encoder output feeds the pooling")
    pooling: str = _f("first", "'first' (CLS) or 'mean' (mask-weighted)")
    # projection head
    add_linear_head: bool = _f(False)
    projection_in_dim: int = _f(768)
    projection_out_dim: int = _f(768)
    dtype: Optional[str] = _f("float32", "float32 | float16 | bfloat16; float16 is the reference's --fp16 format (BERT-family inference runs it natively; T5 and every training path are served in bfloat16)")
    encoder_only: bool = _f(False, "load only the encoder stack of a T5 checkpoint")
    pos_token: Optional[str] = _f(None, "re-ranker: token meaning 'relevant'")
    neg_token: Optional[str] = _f(None, "re-ranker: token meaning 'irrelevant'")
    normalize: bool = _f(False, "L2-normalise the embeddings")


@dataclass
class DataArguments:
    train_dir: str = _f(None, "directory of training files")
    train_path: str = _f(None, "single training file")
    eval_path: str = _f(None, "evaluation file")
    query_path: str = _f(None, "query file")
    corpus_path: str = _f(None, "corpus file")
    data_dir: str = _f(None, "data directory")
    data_path: str = _f(None, "single data file")
    processed_data_path: str = _f(None, "pre-tokenised data directory")
    dataset_name: str = _f(None, "hub dataset name")
    passage_field_separator: str = _f(" ")
    dataset_proc_num: int = _f(12, "pre-processing worker count")
    train_n_passages: int = _f(8)
    positive_passage_no_shuffle: bool = _f(False, "always take the first positive")
    negative_passage_no_shuffle: bool = _f(False, "always take the first negatives")
    encode_in_path: List[str] = _f(None, "files to encode")
    encode_is_qry: bool = _f(False)
    encode_num_shard: int = _f(1)
    encode_shard_index: int = _f(0)
    q_max_len: int = _f(32, "query length after tokenisation (truncate / pad to this)")
    p_max_len: int = _f(128, "passage length after tokenisation (truncate / pad to this)")
    data_cache_dir: Optional[str] = _f(None, "download cache for datasets")
    query_template: str = _f("<text>", "query text template")
    query_column_names: str = _f("id,text", "tsv columns of the query file")
    doc_template: str = _f("Title: <title> Text: <text>", "passage text template")
    doc_column_names: str = _f("id,title,text", "tsv columns of the corpus file")


# `--overwrite_output_dir` is read by the reference's training drivers (train_dr.py:31-39) but
# transformers >= 5 no longer defines it on TrainingArguments; declared below when missing so the
# reference's command lines keep parsing on either version.
_HAS_OVERWRITE = any(f.name == "overwrite_output_dir" for f in __import__("dataclasses").fields(TrainingArguments))


@dataclass
class DRTrainingArguments(TrainingArguments):
    if not _HAS_OVERWRITE:
        overwrite_output_dir: bool = _f(False, "allow a non-empty output_dir")
    warmup_ratio: float = _f(0.1)
    remove_unused_columns: Optional[bool] = _f(False, "keep every dataset column")
    negatives_x_device: bool = _f(False, "use every rank's passages as negatives")
    do_encode: bool = _f(False, "run the encoding loop")
    grad_cache: bool = _f(False, "gradient-cache update")
    gc_q_chunk_size: int = _f(4)
    gc_p_chunk_size: int = _f(32)


@dataclass
class RRTrainingArguments(TrainingArguments):
    if not _HAS_OVERWRITE:
        overwrite_output_dir: bool = _f(False, "allow a non-empty output_dir")
    warmup_ratio: float = _f(0.1)
    remove_unused_columns: Optional[bool] = _f(False, "keep every dataset column")
    margin: float = _f(1.0)
    loss_fn: str = _f("bce", "pair loss name")


@dataclass
class InferenceArguments(TrainingArguments):
    use_gpu: bool = _f(False, "kept for CLI compatibility: search always runs on the MI355X shard")
    encoded_save_path: str = _f(None, "where to write encodings")
    trec_save_path: str = _f(None, "where to write the TREC run")
    trec_run_path: str = _f(None, "first-stage TREC run to re-rank")
    id_key_name: str = _f("id", "name of the id field")
    reranking_depth: int = _f(None, "how many first-stage hits to re-rank")
