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
import ctypes

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

    def __call__(self, center, offset, foreground=None):
        c, s, nc = instance_centers(center, self.threshold, self.kernel, self.top_k,
                                    foreground if self.apply_fg else None)
        ids = instance_assign(offset, c, nc, foreground, self.normalized, self.dist)
        return {'instance_predicted_centers': c, 'instance_predicted_centers_scores': s,
                'instance_predicted_centers_count': nc, 'instance_segmentation_idx': ids}


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
    (N,H,W) int32 from `instance_assign` on the thing pixels -> dict(semantic (-1 = void),
    instance, panoptic id = (class + 1) * label_divisor + instance, 0 = void)"""
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
    return {'semantic': pan_sem, 'instance': pan_inst, 'panoptic': pan_id}


class PanopticPostprocessing:
    """`get_postprocessing_class('panoptic', ...)` of /root/reference/emsanet/decoder.py:141-155:
    semantic arg-max -> thing mask as instance foreground -> centres / grouping -> merge."""

    def __init__(self, instance_postprocessing, semantic_classes_is_thing, label_divisor=1000):
        self.inst = instance_postprocessing
        self.is_thing = tuple(bool(t) for t in semantic_classes_is_thing)
        self.label_divisor = label_divisor

    @property
    def max_instances_per_category(self):
        """the attribute /root/reference/inference_dataset.py:723-724 reads off
        `model.decoders['panoptic_helper'].postprocessing`: panoptic id = (class + 1) * this + instance"""
        return self.label_divisor

    def __call__(self, semantic_logits, center, offset):
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
        return r


# ---------------------------------------------------------------------------------------------
# full-resolution predictions
# ---------------------------------------------------------------------------------------------
_FULLRES_LABEL_KEYS = ('instance_segmentation_idx', 'instance_segmentation_gt_foreground',
                       'panoptic_foreground_mask', 'panoptic_segmentation_deeplab',
                       'panoptic_segmentation_deeplab_semantic_idx',
                       'panoptic_segmentation_deeplab_instance_idx', 'scene_class_idx')


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
    resolution (`emsa_softmax_argmax`); label maps (instance / panoptic ids, masks) are resampled with
    nearest neighbour (`emsa_nearest_fwd_t` on their exact fp32 image: ids < 2^24).  Equal shapes: the
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
    for k in _FULLRES_LABEL_KEYS:
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
