# -*- coding: utf-8 -*-
"""
Eval-time post-processing and input normalisation on device.

The reference wires its post-processing classes into the decoders
(/root/reference/emsanet/decoder.py:61-69,95-104 with the parameters of args.py:468-504); the
classes themselves are part of the un-vendored `nicr_mt_scene_analysis` library, so the instance
grouping below follows the published Panoptic-DeepLab procedure the library implements
(threshold -> max-pool NMS -> top-k centres; every pixel votes for the centre nearest to
pixel + offset) -- restated in oracle/postprocessing_oracle.py, PARITY UNPINNED -- while
arg-max / softmax score are unambiguous.  Arithmetic: csrc/postproc.hip.
"""
import collections.abc
import copy
import ctypes
import math

import torch

from . import _lib
from . import functional as Fn
from ._lib import check


def softmax_argmax(logits):
    """logits (N,C,H,W) or (N,C) -> (score, idx): idx = argmax over C (int64), score = its
    softmax probability (== torch.softmax(logits, 1).max(1))"""
    if logits.dim() == 2:
        n, c = logits.shape
        cp = Fn.pad4(c)
        # rows of `cp` floats, one "pixel" per sample (the row stride is passed as such: with ONE
        # sample no stride of a (1, C, 1, 1) view says what it is -- batch-1 inference)
        x = torch.cat([logits, logits.new_zeros(n, cp - c)], 1) if cp != c else logits.contiguous()
        score = Fn._empty((n,), x.device)
        idx = torch.empty((n,), device=x.device, dtype=torch.int64)
        check(_lib.lib().emsa_softmax_argmax(Fn._p(x), cp, c, n, Fn._p(score), idx.data_ptr(),
                                             Fn._stream()), 'emsa_softmax_argmax')
        return score, idx
    x = Fn.as_act(logits)
    n, c, h, w = x.shape
    ld = Fn.ld_of(x)
    if ld % 4:                   # rows must be 16-byte aligned: re-lay out with a padded stride
        ld = Fn.pad4(c)
        xp = Fn.act_empty(n, ld, h, w, x.device)
        xp[:, :c].copy_(x)
        x = xp[:, :c]
    score = Fn._empty((n, h, w), x.device)
    idx = torch.empty((n, h, w), device=x.device, dtype=torch.int64)
    check(_lib.lib().emsa_softmax_argmax(Fn._p(x), ld, c, n * h * w, Fn._p(score),
                                         idx.data_ptr(), Fn._stream()), 'emsa_softmax_argmax')
    return score, idx


def _u8(m):
    return None if m is None else m.reshape(-1).to(torch.uint8).contiguous()


def instance_centers(heatmap, threshold=0.1, nms_kernel_size=17, top_k=64, foreground=None,
                     return_survivors=False):
    """heatmap (N,1,H,W) -> centers (N,top_k,2) float (y, x; -1 padded), scores (N,top_k),
    n_centers (N,) int32 (defaults: /root/reference/emsanet/args.py:469-504).  The top-k is exact
    over ALL pixels that survive the NMS (ref decoder.py:95-104), also on saturated plateaus with
    tens of thousands of survivors; return_survivors: also that count per image (N,) int32"""
    x = Fn.as_act(heatmap)
    n, _, h, w = x.shape
    L = _lib.lib()
    cap = L.emsa_center_ws_entries(h, w, top_k)
    if cap <= 0 or top_k > L.emsa_center_candidates_max() // 2:
        raise _lib.EmsaError(f"instance_centers: top_k={top_k} outside [1, "
                             f"{L.emsa_center_candidates_max() // 2}]")
    dev = x.device
    ws_count = torch.empty(2 * n, device=dev, dtype=torch.int32)
    ws_score = Fn._empty((n * cap,), dev)
    ws_pos = torch.empty(n * cap, device=dev, dtype=torch.int32)
    centers = Fn._empty((n, top_k, 2), dev)
    scores = Fn._empty((n, top_k), dev)
    n_centers = torch.empty(n, device=dev, dtype=torch.int32)
    fg = _u8(foreground)
    check(L.emsa_instance_centers(Fn._p(x), Fn.ld_of(x), n, h, w, nms_kernel_size, threshold,
                                  top_k, Fn._p(fg), ws_count.data_ptr(), Fn._p(ws_score),
                                  ws_pos.data_ptr(), Fn._p(centers), Fn._p(scores),
                                  n_centers.data_ptr(), Fn._stream()), 'emsa_instance_centers')
    if return_survivors:
        return centers, scores, n_centers, ws_count[n:]
    return centers, scores, n_centers


def instance_assign(offsets, centers, n_centers, foreground=None, normalized_offset=True,
                    offset_distance_threshold=None):
    """offsets (N,2,H,W) (dy, dx; in units of the image height / width when normalized) ->
    instance ids (N,H,W) int32: 1 + index of the nearest centre, 0 = no instance"""
    o = Fn.as_act(offsets)
    n, _, h, w = o.shape
    ids = torch.empty((n, h, w), device=o.device, dtype=torch.int32)
    sy, sx = (float(h), float(w)) if normalized_offset else (1.0, 1.0)
    fg = _u8(foreground)
    check(_lib.lib().emsa_instance_assign(Fn._p(o), Fn.ld_of(o), n, h, w, sy, sx, Fn._p(centers),
                                          n_centers.data_ptr(), centers.shape[1], Fn._p(fg),
                                          float(offset_distance_threshold or 0.0),
                                          ids.data_ptr(), Fn._stream()), 'emsa_instance_assign')
    return ids


class InstancePostprocessing:
    """parameters as `get_postprocessing_class('instance', ...)` receives them
    (/root/reference/emsanet/decoder.py:95-104)"""

    def __init__(self, heatmap_threshold=0.1, heatmap_nms_kernel_size=17,
                 heatmap_apply_foreground_mask=False, top_k_instances=64, normalized_offset=True,
                 offset_distance_threshold=None):
        self.threshold = heatmap_threshold
        self.kernel = heatmap_nms_kernel_size
        self.apply_fg = heatmap_apply_foreground_mask
        self.top_k = top_k_instances
        self.normalized = normalized_offset
        self.dist = offset_distance_threshold

    def __call__(self, center, offset, foreground=None, with_meta=False):
        c, s, nc = instance_centers(center, self.threshold, self.kernel, self.top_k,
                                    foreground if self.apply_fg else None)
        ids = instance_assign(offset, c, nc, foreground, self.normalized, self.dist)
        r = {'instance_predicted_centers': c, 'instance_predicted_centers_scores': s,
             'instance_predicted_centers_count': nc, 'instance_segmentation_idx': ids}
        if with_meta:
            r['instance_segmentation_meta'] = instance_meta(c, s, nc, instance_stats(ids, self.top_k + 1))
        return r


# ImageNet statistics on [0, 1]-scaled RGB: [U] (the library's NormalizeRGB is not vendored)
RGB_MEAN, RGB_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def normalize_rgb(rgb_u8_hwc, mean=RGB_MEAN, std=RGB_STD, scale=1.0 / 255.0):
    """uint8 (N,H,W,3) on the GPU -> float (N,3,H,W)  (`NormalizeRGB` + `ToTorchTensors`,
    /root/reference/emsanet/preprocessing.py:216-226): the H2D copy moves 1 byte per value"""
    x = rgb_u8_hwc.contiguous()
    if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[-1] != 3 or not x.is_cuda:
        raise _lib.EmsaError("normalize_rgb expects a uint8 (N,H,W,3) tensor on the GPU")
    n, h, w, _ = x.shape
    out = Fn._empty((n, 3, h, w), x.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    check(_lib.lib().emsa_normalize_rgb(x.data_ptr(), Fn._p(out), n, h, w, scale, m, s,
                                        Fn._stream()), 'emsa_normalize_rgb')
    return out


def normalize_depth(depth_u16, mean, std, keep_invalid_zero=True):
    """uint16 (N,H,W) depth in mm on the GPU -> float (N,1,H,W): (d - mean)/std, invalid pixels
    (0) stay 0  (`NormalizeDepth(depth_mean, depth_std, raw_depth)`, preprocessing.py:218-224)"""
    x = depth_u16.contiguous()
    if x.dtype not in (torch.uint16, torch.int16) or not x.is_cuda:
        raise _lib.EmsaError("normalize_depth expects a uint16 tensor on the GPU")
    out = Fn._empty((x.shape[0], 1) + tuple(x.shape[-2:]), x.device)
    check(_lib.lib().emsa_normalize_depth(x.data_ptr(), Fn._p(out), x.numel(), float(mean),
                                          float(std), 1 if keep_invalid_zero else 0,
                                          Fn._stream()), 'emsa_normalize_depth')
    return out


_THING_TABLES = {}


def thing_table(classes_is_thing, device):
    """uint8 is-thing lookup table on `device`, created ONCE per (classes, device): building it per
    call was a synchronous pageable host-to-device copy, which stream capture (hipGraph) rejects"""
    key = (tuple(bool(t) for t in classes_is_thing), str(device))
    t = _THING_TABLES.get(key)
    if t is None:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise _lib.EmsaError("is-thing table requested for the first time inside a stream "
                                 "capture: run the post-processing once before capturing")
        t = torch.tensor([1 if b else 0 for b in key[0]], dtype=torch.uint8, device=device)
        _THING_TABLES[key] = t
    return t


def panoptic_merge(semantic_idx, instance_ids, classes_is_thing, top_k=64, label_divisor=1000):
    """Panoptic-DeepLab merge on device.  semantic_idx (N,H,W) int64 in [0, C); instance_ids
    (N,H,W) int32 from `instance_assign` on the thing pixels -> dict(semantic: class in the label
    list WITH void (0 = void, class c -> c + 1: /root/reference/inference_dataset.py:298-304),
    instance, panoptic id = semantic * label_divisor + instance (0 = void), instance_class
    (N, top_k + 1) int32: class WITHOUT void of every instance id, -1 = instance without pixels)"""
    n, h, w = semantic_idx.shape
    dev = semantic_idx.device
    nc = len(classes_is_thing)
    thing = thing_table(classes_is_thing, dev)
    sem = semantic_idx.contiguous()
    ids = instance_ids.contiguous()
    ws_votes = torch.empty(n * (top_k + 1) * nc, device=dev, dtype=torch.int32)
    ws_class = torch.empty(n * (top_k + 1), device=dev, dtype=torch.int32)
    pan_sem = torch.empty((n, h, w), device=dev, dtype=torch.int64)
    pan_inst = torch.empty((n, h, w), device=dev, dtype=torch.int32)
    pan_id = torch.empty((n, h, w), device=dev, dtype=torch.int64)
    check(_lib.lib().emsa_panoptic_merge(sem.data_ptr(), ids.data_ptr(), thing.data_ptr(), n, h * w,
                                         nc, top_k, label_divisor, ws_votes.data_ptr(),
                                         ws_class.data_ptr(), pan_sem.data_ptr(),
                                         pan_inst.data_ptr(), pan_id.data_ptr(), Fn._stream()),
          'emsa_panoptic_merge')
    return {'semantic': pan_sem, 'instance': pan_inst, 'panoptic': pan_id,
            'instance_class': ws_class.view(n, top_k + 1)}


def instance_stats(ids, slots, value=None, mask=None):
    """ids (N,H,W) int32 -> area (N,slots) int32 [, sum (N,slots) int64 of floor(value * 2^30 + .5)]:
    per-instance pixel count (and fixed-point score sum) by integer atomics (order-independent)"""
    n, h, w = ids.shape
    dev = ids.device
    ids = ids.contiguous()
    area = torch.empty((n, slots), device=dev, dtype=torch.int32)
    ssum = torch.empty((n, slots), device=dev, dtype=torch.int64) if value is not None else None
    v = None if value is None else value.contiguous()
    check(_lib.lib().emsa_instance_stats(ids.data_ptr(), Fn._p(v), Fn._p(_u8(mask)), n, h * w, slots,
                                         Fn._p(ssum), area.data_ptr(), Fn._stream()),
          'emsa_instance_stats')
    return (area, ssum) if value is not None else area


def panoptic_scores(semantic_score, pan_instance, pan_semantic, center_scores):
    """score maps of the merged segmentation (`compute_scores=True`,
    /root/reference/emsanet/decoder.py:152).  What the reference's scripts say about them
    (inference_dataset.py:505-517): instance score = "score_instance_center", panoptic score =
    "score_instance_center * (mean_semantic_score_of_instance)".  [U] beyond that (the library is not
    vendored): the mean runs over ALL pixels of the instance and uses each pixel's own arg-max score;
    stuff pixels carry (own semantic score, 0, own semantic score); void pixels 0.
    -> dict(semantic_score, instance_score, panoptic_score (N,H,W) float; per instance (N, top_k+1):
    area int32, semantic_score, panoptic_score float)"""
    n, h, w = pan_instance.shape
    dev = pan_instance.device
    top_k = center_scores.shape[1]
    slots = top_k + 1
    ws_sum = torch.empty((n, slots), device=dev, dtype=torch.int64)
    area = torch.empty((n, slots), device=dev, dtype=torch.int32)
    inst_sem = Fn._empty((n, slots), dev)
    inst_pan = Fn._empty((n, slots), dev)
    o = [Fn._empty((n, h, w), dev) for _ in range(3)]
    check(_lib.lib().emsa_panoptic_scores(Fn._p(semantic_score.contiguous()), pan_instance.data_ptr(),
                                          pan_semantic.data_ptr(), Fn._p(center_scores), n, h * w,
                                          top_k, ws_sum.data_ptr(), area.data_ptr(), Fn._p(inst_sem),
                                          Fn._p(inst_pan), Fn._p(o[0]), Fn._p(o[1]), Fn._p(o[2]),
                                          Fn._stream()), 'emsa_panoptic_scores')
    return {'semantic_score': o[0], 'instance_score': o[1], 'panoptic_score': o[2], 'area': area,
            'instance_semantic_score': inst_sem, 'instance_panoptic_score': inst_pan}


def instance_orientation_sums(orientation, ids, slots, mask=None):
    """orientation (N,2,H,W) (channels: sin, cos -- [U], like oracle/instance_loss_oracle.Spec), ids
    (N,H,W) int32 -> (vec (N,slots,2) int64 fixed-point sums, count (N,slots) int32)"""
    o = Fn.as_act(orientation.float() if orientation.dtype != torch.float32 else orientation)
    n, _, h, w = o.shape
    dev = o.device
    ids = ids.contiguous()
    vec = torch.empty((n, slots, 2), device=dev, dtype=torch.int64)
    cnt = torch.empty((n, slots), device=dev, dtype=torch.int32)
    check(_lib.lib().emsa_instance_orientation(Fn._p(o), Fn.ld_of(o), ids.data_ptr(), Fn._p(_u8(mask)),
                                               n, h * w, slots, vec.data_ptr(), cnt.data_ptr(),
                                               Fn._stream()), 'emsa_instance_orientation')
    return vec, cnt


class Deferred(collections.abc.Sequence):
    """a per-sample list of Python dictionaries (instance meta, orientations) that is built from
    device tensors the first time it is READ: building it needs a device-to-host copy, which a
    captured eval forward (`GraphedInference(..., do_postprocessing=True)`) must not issue.  Under a
    graph the tensors are the static outputs, so `refresh()` (or reading a fresh result) after a
    replay yields the dictionaries of that replay."""

    def __init__(self, build):
        self._build, self._value = build, None

    def refresh(self):
        self._value = None
        return self

    def _get(self):
        if self._value is None:
            self._value = self._build()
        return self._value

    def __len__(self):
        return len(self._get())

    def __getitem__(self, i):
        return self._get()[i]

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return repr(self._get())

    def __deepcopy__(self, memo):                # visualization.py:624,877 deep-copies the meta
        return copy.deepcopy(self._get(), memo)

    def __reduce__(self):
        return (list, (self._get(),))


def _host(*tensors):
    return [t.detach().cpu().numpy() for t in tensors]


def instance_meta(centers, scores, n_centers, area, instance_class=None, semantic_score=None,
                  panoptic_score=None):
    """per sample {instance id: {'center': [y, x], 'score', 'area'[, 'semantic_idx', 'semantic_score',
    'panoptic_score']}} -- every centre found, the last three only for instances that own pixels
    ("filter instances without pixels", /root/reference/inference_dataset.py:420-422,532-533);
    'semantic_idx' WITH void like the map.  Plain ints / floats: the consumer writes it as JSON
    (inference_dataset.py:541-542).  Field names beyond those the scripts read are [U]."""
    def build():
        cen, sc, nc, ar = _host(centers, scores, n_centers, area)
        extra = _host(instance_class, semantic_score, panoptic_score) if instance_class is not None \
            else None
        out = []
        for i in range(cen.shape[0]):
            d = {}
            for k in range(int(nc[i])):
                m = {'center': [float(cen[i, k, 0]), float(cen[i, k, 1])], 'score': float(sc[i, k]),
                     'area': int(ar[i, k + 1])}
                if extra is not None and m['area'] > 0 and extra[0][i, k + 1] >= 0:
                    m['semantic_idx'] = int(extra[0][i, k + 1]) + 1
                    m['semantic_score'] = float(extra[1][i, k + 1])
                    m['panoptic_score'] = float(extra[2][i, k + 1])
                d[k + 1] = m
            out.append(d)
        return out
    return Deferred(build)


TWO_PI = 2.0 * math.pi


def orientation_dicts(vec, count, keep=None):
    """per sample {instance id: angle in [0, 2 pi)} = atan2(sum sin, sum cos) over the instance's
    pixels, for ids with pixels (and keep[sample][id] true).  Consumers:
    /root/reference/emsanet/visualization.py:752-813,905-914.  Range / averaging rule: [U]."""
    def build():
        v, c = _host(vec, count)
        k = None if keep is None else _host(keep)[0]
        out = []
        for i in range(v.shape[0]):
            d = {}
            for j in range(1, v.shape[1]):
                if c[i, j] > 0 and (k is None or k[i, j]):
                    d[j] = math.atan2(float(v[i, j, 0]), float(v[i, j, 1])) % TWO_PI
            out.append(d)
        return out
    return Deferred(build)


def gt_instance_orientations(orientation, batch):
    """'orientations_gt_instance_gt_orientation_foreground': predicted orientation averaged inside the
    GROUND-TRUTH instances (batch['instance']) restricted to batch['orientation_foreground']
    (/root/reference/emsanet/visualization.py:749-765 zips it with batch['instance']) -- or None when
    the batch does not carry both.  The number of id slots is read from the labels (one host sync:
    not available inside a captured forward, where the key is left out)."""
    inst = batch.get('instance') if batch is not None else None
    fg = batch.get('orientation_foreground') if batch is not None else None
    if inst is None or fg is None or torch.cuda.is_current_stream_capturing():
        return None
    if inst.dim() == 4:
        inst = inst[:, 0]
    if fg.dim() == 4:
        fg = fg[:, 0]
    ids = inst.to(torch.int32)
    slots = max(int(ids.max()) + 1, 2)
    vec, cnt = instance_orientation_sums(orientation, ids, slots, fg)
    return orientation_dicts(vec, cnt)


class PanopticPostprocessing:
    """`get_postprocessing_class('panoptic', ...)` of /root/reference/emsanet/decoder.py:141-155:
    semantic arg-max -> thing mask as instance foreground -> centres / grouping -> merge."""

    def __init__(self, instance_postprocessing, semantic_classes_is_thing, label_divisor=1000,
                 semantic_class_has_orientation=None, compute_scores=True):
        self.inst = instance_postprocessing
        self.is_thing = tuple(bool(t) for t in semantic_classes_is_thing)
        self.label_divisor = label_divisor
        self.has_orientation = tuple(bool(t) for t in semantic_class_has_orientation) \
            if semantic_class_has_orientation is not None else (True,) * len(self.is_thing)
        self.compute_scores = compute_scores

    @property
    def max_instances_per_category(self):
        """the attribute /root/reference/inference_dataset.py:723-724 reads off
        `model.decoders['panoptic_helper'].postprocessing`: panoptic id = (class + 1) * this + instance"""
        return self.label_divisor

    def __call__(self, semantic_logits, center, offset, orientation=None):
        score, idx = softmax_argmax(semantic_logits)
        thing = thing_table(self.is_thing, idx.device).bool()[idx]           # (N,H,W) bool
        r = {'semantic_segmentation_score': score, 'semantic_segmentation_idx': idx,
             'panoptic_foreground_mask': thing}
        inst = InstancePostprocessing(self.inst.threshold, self.inst.kernel, True, self.inst.top_k,
                                      self.inst.normalized, self.inst.dist)(center, offset, thing)
        r.update(inst)
        m = panoptic_merge(idx, inst['instance_segmentation_idx'], self.is_thing, self.inst.top_k,
                           self.label_divisor)
        r['panoptic_segmentation_deeplab'] = m['panoptic']
        r['panoptic_segmentation_deeplab_semantic_idx'] = m['semantic']
        r['panoptic_segmentation_deeplab_instance_idx'] = m['instance']
        area = sc = None
        if self.compute_scores:
            sc = panoptic_scores(score, m['instance'], m['semantic'],
                                 inst['instance_predicted_centers_scores'])
            area = sc['area']
            r['panoptic_segmentation_deeplab_semantic_score'] = sc['semantic_score']
            r['panoptic_segmentation_deeplab_instance_score'] = sc['instance_score']
            r['panoptic_segmentation_deeplab_panoptic_score'] = sc['panoptic_score']
        else:
            area = instance_stats(m['instance'], self.inst.top_k + 1)
        r['panoptic_segmentation_deeplab_instance_meta'] = instance_meta(
            inst['instance_predicted_centers'], inst['instance_predicted_centers_scores'],
            inst['instance_predicted_centers_count'], area,
            *((m['instance_class'], sc['instance_semantic_score'], sc['instance_panoptic_score'])
              if sc is not None else ()))
        if orientation is not None:
            # instances of a class that carries orientations (semantic_class_has_orientation,
            # /root/reference/emsanet/decoder.py:151 <- model.py:43)
            vec, cnt = instance_orientation_sums(orientation, m['instance'], self.inst.top_k + 1)
            table = thing_table(self.has_orientation, idx.device)
            cls = m['instance_class']
            keep = table[cls.clamp(min=0).long()].bool() & (cls >= 0)
            r['orientations_panoptic_segmentation_deeplab_instance'] = orientation_dicts(vec, cnt, keep)
        return r


# ---------------------------------------------------------------------------------------------
# full-resolution predictions
# ---------------------------------------------------------------------------------------------
_FULLRES_LABEL_KEYS = ('instance_segmentation_idx', 'instance_segmentation_gt_foreground',
                       'panoptic_foreground_mask', 'panoptic_segmentation_deeplab',
                       'panoptic_segmentation_deeplab_semantic_idx',
                       'panoptic_segmentation_deeplab_instance_idx', 'scene_class_idx')


_FULLRES_SCORE_KEYS = ('panoptic_segmentation_deeplab_semantic_score',
                       'panoptic_segmentation_deeplab_instance_score',
                       'panoptic_segmentation_deeplab_panoptic_score')


def fullres_shape(batch):
    """(H, W) of the full-resolution frames the batch carries (`rgb_fullres` / `depth_fullres`,
    /root/reference/emsanet/tests/test_interface_model.py:86-91), or None"""
    for k in ('rgb_fullres', 'depth_fullres', 'rgbd_fullres'):
        t = batch.get(k) if batch is not None else None
        if torch.is_tensor(t) and t.dim() >= 2:
            return int(t.shape[-2]), int(t.shape[-1])
    return None


def add_fullres_predictions(r, hw):
    """`<key>_fullres` entries of the post-processed dict -- what the reference's consumers read through
    `get_fullres(prediction, key)` (/root/reference/inference_dataset.py:223-225,284,298,411,468-520;
    inference_samples.py:153-163): predictions at the resolution of the un-resized input frames.
    How the un-vendored library resamples is [U]; here: the semantic LOGITS are up-sampled bilinearly
    (align_corners=False, `emsa_bilinear_fwd_t`) and arg-max / softmax score are taken at full
    resolution (`emsa_softmax_argmax`); label maps (instance / panoptic ids, masks) and the per-segment
    score maps are resampled with nearest neighbour (`emsa_nearest_fwd_t` on their exact fp32 image:
    ids < 2^24).  Equal shapes: the
    plain entries are aliased."""
    hf, wf = hw
    out = {}
    logits = r.get('semantic_output')
    if logits is not None and 'semantic_segmentation_idx' in r:
        n, c, h, w = logits.shape
        if (h, w) == (hf, wf):
            out['semantic_segmentation_idx_fullres'] = r['semantic_segmentation_idx']
            out['semantic_segmentation_score_fullres'] = r['semantic_segmentation_score']
        else:
            x = Fn.as_act(logits.float() if logits.dtype != torch.float32 else logits, dense=True)
            big = Fn.act_empty(n, c, hf, wf, x.device)
            Fn.bilinear_fwd(x, big)
            score, idx = softmax_argmax(big)
            out['semantic_segmentation_idx_fullres'] = idx
            out['semantic_segmentation_score_fullres'] = score
    for k in _FULLRES_LABEL_KEYS + _FULLRES_SCORE_KEYS:
        t = r.get(k)
        if not torch.is_tensor(t) or t.dim() != 3:
            continue
        n, h, w = t.shape
        if (h, w) == (hf, wf):
            out[k + '_fullres'] = t
            continue
        x = Fn.as_act(t.to(torch.float32).unsqueeze(1), dense=True)
        big = Fn.act_empty(n, 1, hf, wf, x.device)
        Fn.nearest_fwd(x, big)
        out[k + '_fullres'] = big[:, 0].to(t.dtype)
    r.update(out)
    return r
