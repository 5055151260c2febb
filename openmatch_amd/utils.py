"""Run-file I/O, templates and result merging with the reference's names and behaviour
(src/openmatch/utils.py:126-235).  Pure host logic: these touch Python dicts and text files,
not tensors; the tensor-side counterparts (`om_topk_merge`, pooling inside
`om_encoder_forward`) live in the HIP library."""
import warnings
from typing import Dict, List

import torch


def save_as_trec(rank_result: Dict[str, Dict[str, float]], output_path: str, run_id: str = "OpenMatch"):
    """One line per hit: `<qid> Q0 <docid> <rank> <score> <run_id>`, hits ordered by score (desc)."""
    with open(output_path, "w") as out:
        for qid, hits in rank_result.items():
            ordered = sorted(hits.items(), key=lambda kv: kv[1], reverse=True)
            out.writelines(f"{qid} Q0 {doc} {rank} {score} {run_id}\n"
                           for rank, (doc, score) in enumerate(ordered, start=1))


def load_from_trec(input_path: str, as_list: bool = False, max_len_per_q: int = None):
    """Reads 6-column TREC runs or 3-column `<qid> <docid> <score>` files."""
    result, seen = {}, 0
    with open(input_path) as src:
        for line in src:
            cols = line.split()
            if len(cols) == 6:
                qid, doc, score = cols[0], cols[2], cols[4]
            elif len(cols) == 3:
                qid, doc, score = cols
            else:
                raise ValueError("Invalid run format")
            if qid not in result:
                result[qid] = [] if as_list else {}
                seen = 0
            if max_len_per_q is None or seen < max_len_per_q:
                if as_list:
                    result[qid].append((doc, float(score)))
                else:
                    result[qid][doc] = float(score)
            seen += 1
    return result


def find_all_markers(template: str):
    """Names between '<' and '>' in order of appearance."""
    names, pos = [], 0
    while True:
        lo = template.find("<", pos)
        hi = template.find(">", lo) if lo != -1 else -1
        if lo == -1 or hi == -1:
            return names
        names.append(template[lo + 1:hi])
        pos = hi + 1


def fill_template(template: str, data: Dict, markers: List[str] = None, allow_not_found: bool = False):
    """Substitute every `<a.b>` marker by data['a']['b']."""
    for marker in (find_all_markers(template) if markers is None else markers):
        value = data
        for part in marker.split("."):
            value = value.get(part, None)
            if value is None:
                break
        if value is None:
            if not allow_not_found:
                raise ValueError("Cannot find the marker '{}' in the data".format(marker))
            warnings.warn("Marker '{}' not found in data. Replacing it with an empty string.".format(marker),
                          RuntimeWarning)
            value = ""
        template = template.replace("<{}>".format(marker), str(value))
    return template


def merge_retrieval_results_by_score(results: List[Dict[str, Dict[str, float]]], topk: int = 100):
    """Union of per-partition hit dicts (an id already present keeps its first score), then the
    `topk` best per query; ties keep insertion order (stable sort)."""
    merged: Dict[str, Dict[str, float]] = {}
    for part in results:
        for qid, hits in part.items():
            into = merged.setdefault(qid, {})
            for doc, score in hits.items():
                into.setdefault(doc, score)
    return {qid: dict(sorted(hits.items(), key=lambda kv: kv[1], reverse=True)[:topk])
            for qid, hits in merged.items()}


def mean_pooling(token_embeddings, attention_mask):
    """Mask-weighted mean over the sequence axis; empty masks divide by 1e-9 (reference :233-235)."""
    mask = attention_mask.unsqueeze(-1).expand(token_embeddings.size()).float()
    return torch.sum(token_embeddings * mask, 1) / torch.clamp(mask.sum(1), min=1e-9)


def eval_mrr(qrel, run, cutoff=None):
    """MRR@cutoff as scripts/evaluate.py:5-28 defines it."""
    total, ranked_queries, per_query = 0.0, 0, {}
    for qid, rels in qrel.items():
        if qid not in run:
            continue
        ranked_queries += 1
        order = sorted(run[qid].items(), key=lambda kv: kv[1], reverse=True)
        rr = 0.0
        for rank, (doc, _) in enumerate(order):
            if (cutoff is None or rank < cutoff) and rels.get(doc, 0) > 0:
                rr = 1.0 / (rank + 1)
                break
        per_query[qid] = rr
        total += rr
    per_query["all"] = total / ranked_queries
    return per_query


def eval_ndcg(qrel, run, cutoff=10):
    """nDCG@cutoff the way trec_eval / pytrec_eval's `ndcg_cut` computes it (what the reference's BEIR
    driver asks pytrec_eval for, driver/retrieve_beir.py:63-65): gain = the relevance grade, discount
    1/log2(rank+1), documents ranked by descending score with ties broken by descending doc id,
    ideal ranking from the judged documents; queries without judged relevant documents score 0 and
    only judged queries count.  Returns {qid: value, ..., "all": mean}."""
    import math
    out = {}
    for qid, judged in qrel.items():
        if qid not in run:
            continue
        ranking = sorted(run[qid].items(), key=lambda kv: (kv[1], kv[0]), reverse=True)[:cutoff]
        dcg = sum(max(judged.get(doc, 0), 0) / math.log2(i + 2) for i, (doc, _) in enumerate(ranking))
        ideal = sorted((r for r in judged.values() if r > 0), reverse=True)[:cutoff]
        idcg = sum(r / math.log2(i + 2) for i, r in enumerate(ideal))
        out[qid] = dcg / idcg if idcg > 0 else 0.0
    out["all"] = sum(out.values()) / len(out) if out else 0.0
    return out


def __getattr__(name):
    # `from openmatch.utils import SimpleTrainPreProcessor / SimpleCollectionPreProcessor` (reference
    # utils.py:14,104) -- they live in preprocess.py
    if name in ("SimpleTrainPreProcessor", "SimpleCollectionPreProcessor"):
        from . import preprocess
        return getattr(preprocess, name)
    raise AttributeError(name)
