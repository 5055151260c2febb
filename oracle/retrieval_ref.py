"""Restatement of the Python that surrounds the index on the reference's retrieval path, and
of its loss.  Test infrastructure only — see oracle/__init__.py."""
import numpy as np
import torch
import torch.nn.functional as F


def search_to_dict(D, I, doc_lookup, query_lookup):
    """Retriever.search tail (src/openmatch/retriever/dense_retriever.py:180-190), incl. the
    quirk that I == -1 maps to the LAST doc (`np.array(doc_lookup)[I]`, :181)."""
    original = np.array(doc_lookup)[I]
    out = {}
    for q, (scores, docs) in enumerate(zip(D, original)):
        qid = str(query_lookup[q])
        out[qid] = {}
        for doc, score in zip(docs, scores):
            out[qid][str(doc)] = float(score)
    return out


def merge_retrieval_results_by_score(results, topk=100):
    """src/openmatch/utils.py:215-229: first occurrence of a docid wins, stable sort desc."""
    merged = {}
    for result in results:
        for qid in result:
            merged.setdefault(qid, {})
            for doc_id in result[qid]:
                if doc_id not in merged[qid]:
                    merged[qid][doc_id] = result[qid][doc_id]
    for qid in merged:
        merged[qid] = {k: v for k, v in
                       sorted(merged[qid].items(), key=lambda x: x[1], reverse=True)[:topk]}
    return merged


def trec_lines(rank_result, run_id="OpenMatch"):
    """src/openmatch/utils.py:126-136 save_as_trec (returns the lines instead of writing)."""
    lines = []
    for qid in rank_result:
        ranked = sorted(rank_result[qid].items(), key=lambda x: x[1], reverse=True)
        for i, (doc_id, score) in enumerate(ranked):
            lines.append("{} Q0 {} {} {} {}\n".format(qid, doc_id, i + 1, score, run_id))
    return lines


def eval_mrr(qrel, run, cutoff=None):
    """scripts/evaluate.py:5-28 (the file itself imports pytrec_eval at :1 and cannot load)."""
    mrr, num_ranked_q, results = 0.0, 0, {}
    for qid in qrel:
        if qid not in run:
            continue
        num_ranked_q += 1
        ranked = sorted(run[qid].items(), key=lambda x: x[1], reverse=True)
        rr = 0.0
        for i, (docid, _) in enumerate(ranked):
            if cutoff is None or i < cutoff:
                if docid in qrel[qid] and qrel[qid][docid] > 0:
                    rr = 1.0 / (i + 1)
                    break
        results[qid] = rr
        mrr += rr
    results["all"] = mrr / num_ranked_q
    return results


def contrastive_loss(q_reps, p_reps, n_psg, scale=1.0):
    """DRModel.forward tail (modeling/dense_retrieval_model.py:113-125) / SimpleContrastiveLoss
    (loss.py:9-15): CE(mean) of q@p.T against target i*n_psg, times `scale` (= world_size when
    training with negatives_x_device)."""
    scores = q_reps @ p_reps.t()
    target = torch.arange(scores.size(0), dtype=torch.long) * n_psg
    return F.cross_entropy(scores, target, reduction="mean") * scale, scores


def gather_with_local_grad(parts, rank):
    """dist_gather_tensor (:247-258): all ranks' tensors concatenated rank-major; only the local
    slot carries gradient."""
    return torch.cat([p if r == rank else p.detach() for r, p in enumerate(parts)], 0)
