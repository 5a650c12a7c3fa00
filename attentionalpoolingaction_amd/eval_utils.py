"""Eval consumers of the head's logits.

Reference: /root/reference/src/eval.py:193-197,303-306 (argmax / softmax / accuracy),
/root/reference/src/eval/utils.py:4-16 (`compute_map`) and
/root/reference/src/eval/cap_eval_utils.py:55-109 (`calc_pr_ovr_noref`, `voc_ap`).

`predict` runs the fused HIP softmax/argmax kernel on the GPU; the mAP itself is host-side numpy
like the reference (it runs once per evaluation over [num_samples, K] scores), vectorised but with
the reference's exact ordering rule: scores are ranked by `np.argsort(out)[::-1]`, i.e. ties are
broken by the reversed quicksort order of numpy -- we call the very same numpy routine so the
ranking (and therefore AP under ties) is identical.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def predict(logits):
    """(probs [N,K] float32, predictions [N] int64) on the device of `logits`
    (eval.py:193-197: tf.argmax(logits,1), tf.nn.softmax(logits,-1))."""
    import torch
    from .custom_ops import custom_ops_factory as cof
    labels = torch.zeros((logits.shape[0],), dtype=torch.int64, device=logits.device)
    _, _, probs, pred = cof.softmax_xent_fwd_bwd(logits.contiguous(), labels, want_grad=False,
                                                 want_probs=True, want_pred=True)
    return probs, pred


def voc_ap(rec: np.ndarray, prec: np.ndarray) -> float:
    """Area under the right-max envelope of precision over the recall steps."""
    mrec = np.concatenate(([0.0], np.asarray(rec, dtype=np.float64).ravel(), [1.0]))
    mpre = np.concatenate(([0.0], np.asarray(prec, dtype=np.float64).ravel(), [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    idx = np.nonzero(mrec[1:] != mrec[:-1])[0] + 1
    # accumulate left to right like the reference's python loop (same float64 rounding order)
    ap = 0.0
    for d in (mrec[idx] - mrec[idx - 1]) * mpre[idx]:
        ap = ap + d
    return float(ap)


def calc_pr_ovr_noref(counts: np.ndarray, out: np.ndarray):
    """Precision / recall / AP of one class: counts > 0 marks the positives, `out` the scores."""
    counts = np.array(np.asarray(counts) > 0, dtype=np.float32)
    out = np.asarray(out)
    ind = np.argsort(out)[::-1]
    score = out[ind].astype(np.float64)
    tp = counts[ind].astype(np.float64)
    fp = 1.0 - tp
    ctp = np.cumsum(tp)
    P = ctp / (ctp + np.cumsum(fp))
    R = ctp / np.sum(counts)
    return P, R, score, voc_ap(R, P)


def compute_map(all_logits: np.ndarray, all_labels: np.ndarray) -> Tuple[float, List[float]]:
    """mAP over the classes that have at least one positive (eval/utils.py:4-16)."""
    all_logits = np.asarray(all_logits)
    all_labels = np.asarray(all_labels)
    aps: List[float] = []
    for cid in range(all_logits.shape[1]):
        this_labels = (all_labels == cid).astype('float32')
        if np.sum(this_labels) == 0:
            continue        # the reference prints a notice and skips the class
        aps.append(calc_pr_ovr_noref(this_labels, all_logits[:, cid])[3])
    return float(np.mean(aps)), aps


def accuracy(all_scores: np.ndarray, all_labels: np.ndarray) -> float:
    """eval.py:304-305: mean(argmax(scores, 1) == labels)."""
    return float(np.mean(np.asarray(all_scores).argmax(axis=1) == np.asarray(all_labels)))
