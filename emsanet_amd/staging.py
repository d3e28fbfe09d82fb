"""Host-to-device input staging overlapped with the previous training / inference step.

The model boundary takes device tensors (`EMSANet.forward(batch)`); the reference's loop moves every
batch synchronously right before the step (`/root/reference/main.py:112-123`, timing protocol of
`/root/reference/inference_time_whole_model.py:297-347` includes that copy).  `BatchStager` is the
MI355X-side replacement for that copy:

  * the loader hands over RAW frames in pinned host memory -- uint8 RGB (N,H,W,3) and uint16 depth
    (N,H,W), 1-2 bytes per value instead of the 4 of the normalised float tensors (a bs=32 640x480
    RGB-D batch is 49 MB instead of 157 MB: 0.8 ms instead of 2.5 ms over PCIe);
  * batch i+1 is copied on a separate HIP stream into a ring of device buffers while step i runs;
  * `NormalizeRGB` / `NormalizeDepth` + HWC->CHW (`/root/reference/emsanet/preprocessing.py:216-226`)
    run as kernels on the compute stream (`postprocessing.normalize_*`) when the batch is handed
    out, so nothing but events crosses between the streams.
Float tensors in the incoming dict (already normalised inputs, targets) are staged as they are.
"""
from typing import Dict, Iterable, Iterator

import torch

from . import _lib
from .postprocessing import normalize_depth, normalize_rgb


class BatchStager:
    """iterate over `host_batches` (dicts of pinned CPU tensors), yield dicts of device tensors.

    depth_stats = (mean, std) of the depth normalisation; `ring` device buffer sets: a set is
    refilled (on the copy stream) as soon as the kernels that read its previous content have been
    enqueued.  Non-tensor entries of a batch are passed through."""

    def __init__(self, host_batches: Iterable[Dict], device, depth_stats=(0.0, 1.0), ring: int = 2,
                 rgb_key: str = 'rgb', depth_key: str = 'depth'):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.EmsaError(f"BatchStager stages batches onto an AMD GPU, not onto {self.device}")
        if ring < 2:
            raise ValueError("ring must be >= 2 (one buffer set being drained, one being filled)")
        self.src = host_batches
        self.depth_stats = depth_stats
        self.ring = ring
        self.rgb_key, self.depth_key = rgb_key, depth_key
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._slots = [dict(bufs={}, copied=torch.cuda.Event(), freed=None) for _ in range(ring)]
        self.bytes_staged = 0
        self.batches_staged = 0

    # -- one slot: raw device buffers that the copy stream fills and the compute stream drains ----
    def _issue(self, slot, host_batch):
        with torch.cuda.stream(self.copy_stream):
            if slot['freed'] is not None:
                # the buffers' previous content was consumed by kernels on the compute stream
                self.copy_stream.wait_event(slot['freed'])
            for k, v in host_batch.items():
                if not torch.is_tensor(v):
                    continue
                if v.is_cuda:
                    slot['bufs'][k] = v
                    continue
                if not v.is_pinned():
                    raise _lib.EmsaError(f"batch['{k}'] is pageable host memory: the overlapped copy "
                                         "needs pinned memory (DataLoader(pin_memory=True))")
                buf = slot['bufs'].get(k)
                if buf is None or buf.shape != v.shape or buf.dtype != v.dtype:
                    buf = torch.empty(v.shape, dtype=v.dtype, device=self.device)
                    slot['bufs'][k] = buf
                buf.copy_(v, non_blocking=True)
                self.bytes_staged += v.numel() * v.element_size()
            slot['copied'].record(self.copy_stream)
        slot['extra'] = {k: v for k, v in host_batch.items() if not torch.is_tensor(v)}
        slot['keys'] = [k for k, v in host_batch.items() if torch.is_tensor(v)]
        self.batches_staged += 1

    def _hand_out(self, slot):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(slot['copied'])
        out = dict(slot['extra'])
        for k in slot['keys']:
            raw = slot['bufs'][k]
            if k == self.rgb_key and raw.dtype == torch.uint8:
                out[k] = normalize_rgb(raw)
            elif k == self.depth_key and raw.dtype in (torch.uint16, torch.int16):
                out[k] = normalize_depth(raw, *self.depth_stats)
            else:
                # handed out as is: it must not be overwritten while the step uses it
                out[k] = raw.clone()
        ev = torch.cuda.Event()
        ev.record(cur)           # every reader of the raw buffers has been enqueued before this
        slot['freed'] = ev
        return out

    def __iter__(self) -> Iterator[Dict]:
        it = iter(self.src)
        pending = []                      # slots whose copies are in flight, oldest first
        free = list(self._slots)
        exhausted = False

        def top_up():
            nonlocal exhausted
            while free and not exhausted:
                try:
                    hb = next(it)
                except StopIteration:
                    exhausted = True
                    return
                s = free.pop(0)
                self._issue(s, hb)
                pending.append(s)

        top_up()
        while pending:
            s = pending.pop(0)
            batch = self._hand_out(s)
            free.append(s)
            top_up()                      # the next copy starts before the step's kernels are launched
            yield batch


def pinned_raw_batch(n: int, h: int, w: int, seed: int = 0, modalities=('rgb', 'depth')) -> Dict:
    """synthetic RAW frames in pinned host memory (uint8 RGB, uint16 depth in mm with ~5 % invalid
    pixels), the form a camera driver / decoded dataset sample has before preprocessing"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    if 'rgb' in modalities:
        out['rgb'] = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8).pin_memory()
    if 'depth' in modalities:
        d = torch.randint(300, 8000, (n, h, w), generator=g, dtype=torch.int32)
        d[torch.rand((n, h, w), generator=g) < 0.05] = 0
        out['depth'] = d.to(torch.uint16).pin_memory()
    return out
