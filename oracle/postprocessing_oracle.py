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
    keep their class, thing pixels without an instance become void (-1 / panoptic id 0)"""
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
    pan = torch.where(sem_o < 0, torch.zeros_like(sem_o), (sem_o + 1) * label_divisor + inst_o.long())
    return {'semantic': sem_o, 'instance': inst_o, 'panoptic': pan}
