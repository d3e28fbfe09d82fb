# -*- coding: utf-8 -*-
"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (instance grouping) / torch-defined
(arg-max, softmax score, normalisation arithmetic).

Plain-PyTorch CPU restatement of the eval post-processing the reference configures in
/root/reference/emsanet/decoder.py:95-104 (parameters: args.py:468-504: heatmap threshold 0.1,
NMS kernel 17, top-k 64, normalized offsets for the 'tanh' encoding: preprocessing.py:193-197).
The post-processing classes live in the un-vendored nicr_mt_scene_analysis v0.3.1; the procedure
is the published Panoptic-DeepLab instance grouping (Cheng et al., CVPR 2020, Sec. 3.3):
  1. centres = pixels with heat >= threshold that equal the k x k max-pooled heatmap, the top_k
     by score (ties: lower flattened position first  [U]);
  2. each (foreground) pixel is assigned to the centre closest to pixel + offset, offsets in
     units of the image size when `normalized_offset` [U].
"""
import torch
import torch.nn.functional as F


def softmax_argmax(logits):
    score, idx = torch.softmax(logits.double(), dim=1).max(dim=1)
    return score, idx


def instance_centers(heat, threshold=0.1, kernel=17, top_k=64, foreground=None):
    """heat (N,1,H,W) -> list over images of (centers (k,2) [y,x], scores (k,))"""
    n, _, h, w = heat.shape
    x = heat.clone()
    if foreground is not None:
        x = x * foreground.reshape(x.shape).to(x.dtype)
    pooled = F.max_pool2d(x, kernel, stride=1, padding=kernel // 2)
    keep = (x >= threshold) & (x == pooled)
    out = []
    for i in range(n):
        pos = torch.nonzero(keep[i, 0].reshape(-1)).reshape(-1)
        sc = x[i, 0].reshape(-1)[pos]
        # score descending, position ascending
        order = sorted(range(len(pos)), key=lambda j: (-float(sc[j]), int(pos[j])))[:top_k]
        p = pos[order]
        out.append((torch.stack([p // w, p % w], 1).float(), sc[order]))
    return out


def instance_assign(offsets, centers_per_image, foreground=None, normalized_offset=True,
                    max_distance=None):
    n, _, h, w = offsets.shape
    ids = torch.zeros(n, h, w, dtype=torch.int32)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing='ij')
    sy, sx = (float(h), float(w)) if normalized_offset else (1.0, 1.0)
    for i in range(n):
        c = centers_per_image[i][0]
        if len(c) == 0:
            continue
        cy = yy + offsets[i, 0] * sy
        cx = xx + offsets[i, 1] * sx
        d = (cy[..., None] - c[:, 0]) ** 2 + (cx[..., None] - c[:, 1]) ** 2
        best, arg = d.min(-1)
        lab = (arg + 1).to(torch.int32)
        if max_distance:
            lab[best > max_distance ** 2] = 0
        if foreground is not None:
            lab[~foreground[i].bool()] = 0
        ids[i] = lab
    return ids


def panoptic_merge(semantic_idx, instance_ids, classes_is_thing, label_divisor=1000):
    """every instance takes the majority class (first maximum) of its thing pixels, stuff pixels
    keep their class, thing pixels without an instance become void.  'semantic' is returned in the
    label list WITH void (0 = void, class c -> c + 1), the convention the reference's consumers state
    for 'panoptic_segmentation_deeplab_semantic_idx' (/root/reference/inference_dataset.py:298-304
    "already has void class"; emsanet/visualization.py:726-727 "both with void"); panoptic id =
    semantic * label_divisor + instance, void 0 (emsanet/tests/test_metrics_with_model.py:113-131)"""
    n = semantic_idx.shape[0]
    thing_c = torch.tensor(classes_is_thing, dtype=torch.bool)
    sem_o = torch.full_like(semantic_idx, -1)
    inst_o = torch.zeros_like(instance_ids)
    nc = len(classes_is_thing)
    for i in range(n):
        sem, ids = semantic_idx[i], instance_ids[i]
        thing = thing_c[sem]
        sem_o[i][~thing] = sem[~thing]
        for k in ids.unique().tolist():
            if k <= 0:
                continue
            m = (ids == k) & thing
            if not m.any():
                continue
            votes = torch.bincount(sem[m], minlength=nc)
            cls = int(votes.argmax())            # torch.argmax: first maximum
            sem_o[i][m] = cls
            inst_o[i][m] = k
    sem_o = sem_o + 1
    pan = torch.where(sem_o == 0, torch.zeros_like(sem_o), sem_o * label_divisor + inst_o.long())
    return {'semantic': sem_o, 'instance': inst_o, 'panoptic': pan}


def panoptic_scores(semantic_score, pan_instance, pan_semantic, center_scores):
    """`compute_scores=True` (/root/reference/emsanet/decoder.py:152).  Stated by the reference's
    scripts (inference_dataset.py:505-517): instance score = "score_instance_center", panoptic score =
    "score_instance_center * (mean_semantic_score_of_instance)".  [U]: mean over all pixels of the
    instance of their arg-max score; stuff pixels (own score, 0, own score); void 0.
    The mean is taken over scores rounded to 2^-30 (floor(s * 2^30 + 0.5), exact integer sum) -- the
    rule that makes the result independent of the summation order; it moves a mean by < 1e-9.
    semantic_score (N,H,W) float32, center_scores: list of per-image 1-D float32 tensors
    -> (semantic, instance, panoptic) score maps float32 + per-image dict id -> (area, mean, pan)"""
    import numpy as np
    n = pan_instance.shape[0]
    o_sem = torch.zeros_like(semantic_score)
    o_inst = torch.zeros_like(semantic_score)
    o_pan = torch.zeros_like(semantic_score)
    per_instance = []
    for i in range(n):
        ids = pan_instance[i]
        stuff = (ids == 0) & (pan_semantic[i] > 0)
        o_sem[i][stuff] = semantic_score[i][stuff]
        o_pan[i][stuff] = semantic_score[i][stuff]
        d = {}
        for k in ids.unique().tolist():
            if k <= 0:
                continue
            m = ids == k
            v = semantic_score[i][m].numpy().astype(np.float64)
            v = np.clip(np.nan_to_num(v, nan=0.0), 0.0, 1.0)
            fixed = np.floor(v * 2.0 ** 30 + 0.5).astype(np.int64).sum()
            area = int(m.sum())
            mean = np.float32(float(fixed) / float(area) / 2.0 ** 30)
            cs = np.float32(center_scores[i][k - 1])
            pan = np.float32(cs * mean)
            o_sem[i][m] = float(mean)
            o_inst[i][m] = float(cs)
            o_pan[i][m] = float(pan)
            d[k] = (area, float(mean), float(pan))
        per_instance.append(d)
    return o_sem, o_inst, o_pan, per_instance


def instance_orientations(orientation, ids, mask=None):
    """{id: angle in [0, 2 pi)} per image: atan2(sum sin, sum cos) over the instance's (masked)
    pixels; orientation (N,2,H,W) float32 with channels (sin, cos) [U]; components rounded to 2^-24
    before the (then exact) summation, as the device kernel does"""
    import math
    import numpy as np
    out = []
    for i in range(ids.shape[0]):
        d = {}
        for k in ids[i].unique().tolist():
            if k <= 0:
                continue
            m = ids[i] == k
            if mask is not None:
                m = m & mask[i].bool()
            if not m.any():
                continue
            v = orientation[i][:, m].numpy().astype(np.float64)
            v = np.clip(np.nan_to_num(v, nan=0.0), -16384.0, 16384.0)
            fx = np.rint(v * 2.0 ** 24).astype(np.int64).sum(axis=1)
            d[int(k)] = math.atan2(float(fx[0]), float(fx[1])) % (2.0 * math.pi)
        out.append(d)
    return out
