"""Post-processing of the sliding-window scores: per-frame aggregation of overlapping
windows, thresholded chunks, 3-camera merge, submission writer, F1.  Pure numpy on the host, restated from
scripts/aicity_inf_graph.py:221-351 and scripts/aicity_inf.py:36-129 (vectorised where the result is identical);
pinned by tests/golden/postprocess.json generated from the reference's own functions."""
from collections import defaultdict

import numpy as np


def aggregate_predictions(pred_list, aggregate_func, num_class):
    """aicity_inf_graph.py:313-351: [frame_num, num_class]; frames covered by no window keep the zero padding row."""
    t0s = [t[0] for t in pred_list]
    t1s = [t[1] for t in pred_list]
    lo, hi = min(t0s + t1s), max(t0s + t1s)
    n = hi - lo
    per_frame = [[] for _ in range(n)]
    for t0, t1, score in pred_list:
        assert len(score) == num_class
        for t in range(t0, t1):
            per_frame[t - lo].append(score)
    zero = np.zeros((num_class,), dtype="float32")
    rows = [aggregate_func(np.vstack(s if s else [zero]), axis=0) for s in per_frame]
    return np.vstack(rows)


def get_chunks(score_list, threshold):
    """aicity_inf_graph.py:288-309 (including its quirk: a run that reaches the last frame is closed only if it started
    earlier; a run starting AT the last frame is dropped)."""
    chunks, start = [], None
    n = len(score_list)
    for fidx in range(n):
        if score_list[fidx] >= threshold:
            if start is None:
                start = fidx
            elif fidx == n - 1:
                chunks.append((start, fidx, fidx - start + 1, np.mean(score_list[start:fidx + 1]), score_list[start:fidx + 1]))
                start = None
        elif start is not None:
            chunks.append((start, fidx, fidx - start + 1, np.mean(score_list[start:fidx + 1]), score_list[start:fidx + 1]))
            start = None
    return chunks


def video_action_chunks(preds, action_id_to_thres, video_fps=30.0, use_num_chunk=1, sort_base="score"):
    """aicity_inf.py:74-103: per action the top chunk(s) of one camera file: (start_s, end_s, num_frame, mean_score)."""
    inst = defaultdict(list)
    for action_id, thr in action_id_to_thres.items():
        chunks = get_chunks(preds[:, action_id], thr)
        if not chunks:
            continue
        chunks.sort(key=(lambda x: x[2]) if sort_base == "length" else (lambda x: x[3]), reverse=True)
        for c in chunks[:use_num_chunk]:
            inst[action_id].append((c[0] / video_fps, c[1] / video_fps, c[2], c[3]))
    return inst


def merge_views(test_vids, action_chunks, action_ids, use_num_chunk=1, sort_base="length"):
    """aicity_inf.py:105-126: merge the three synchronised camera files of a video id; (vid, action, start, end) rows
    with the reference's round(start)+1 / round(end)-1 tightening."""
    outputs = []
    for vid, files in test_vids.items():
        for action_id in action_ids:
            allc = [one for f in files if action_id in action_chunks[f] for one in action_chunks[f][action_id]]
            if not allc:
                continue
            allc.sort(key=(lambda x: x[2]) if sort_base == "length" else (lambda x: x[3]), reverse=True)
            for c in allc[:use_num_chunk]:
                outputs.append((vid, action_id, round(c[0]) + 1.0, round(c[1]) - 1.0))
    return outputs


def write_submission(outputs, path):
    """aicity_inf.py:128-131."""
    with open(path, "w") as f:
        for vid, action_id, start, end in outputs:
            f.writelines("%s %s %.6f %.6f\n" % (vid, action_id, start, end))


def compute_f1(anno_data, classes, action_chunks, use_num_chunk=1, sec_thres=1.0, chunk_sort_base="length", return_pr=False,
               use_tight_times=False, use_ori_times=False):
    """aicity_inf_graph.py:221-286."""
    TP = FP = FN = 0
    for vid in anno_data:
        for action_id in classes:
            anno = [o for o in anno_data[vid] if o[-1] == action_id]
            if len(anno) != 3:
                continue
            allc = [one for o in anno if action_id in action_chunks[o[0]] for one in action_chunks[o[0]][action_id]]
            if not allc:
                FN += 1
                continue
            allc.sort(key=(lambda x: x[2]) if chunk_sort_base == "length" else (lambda x: x[3]), reverse=True)
            match_gt = 0
            for c in allc[:use_num_chunk]:
                if use_tight_times:
                    ps, pe = round(c[0]) + 1.0, round(c[1]) - 1.0
                else:
                    ps, pe = round(c[0]), round(c[1])
                if use_ori_times:
                    ps, pe = c[0], c[1]
                gs, ge = anno[0][2], anno[0][3]
                if gs - sec_thres <= ps <= gs + sec_thres and ge - sec_thres <= pe <= ge + sec_thres:
                    if match_gt == 1:
                        FP += 1
                    else:
                        TP += 1
                        match_gt += 1
                else:
                    FP += 1
            if not match_gt:
                FN += 1
    f1 = TP / (TP + 0.5 * (FP + FN))
    if return_pr:
        return f1, TP / (TP + FP), TP / (TP + FN)
    return f1
