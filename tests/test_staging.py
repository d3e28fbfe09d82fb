"""Overlapped host-to-device input staging (emsanet_amd/staging.py): the replacement for the
reference's synchronous per-step copy (/root/reference/main.py:112-123)."""
import pytest
import torch

from util import DEV


def test_stager_refuses_cpu_target():
    from emsanet_amd import _lib
    from emsanet_amd.staging import BatchStager
    with pytest.raises(_lib.EmsaError):
        BatchStager([], 'cpu')


@pytest.mark.gpu
def test_stager_order_content_and_buffer_reuse():
    """7 batches through a ring of 2 buffer sets while a long kernel keeps the compute stream busy:
    every batch arrives in order, normalised exactly like a direct call on its own frames (a buffer
    refilled too early would show up as the NEXT batch's content)"""
    from emsanet_amd.postprocessing import normalize_depth, normalize_rgb
    from emsanet_amd.staging import BatchStager, pinned_raw_batch
    n, h, w = 2, 64, 96
    raws = []
    for i in range(7):
        b = pinned_raw_batch(n, h, w, seed=i)
        b['label'] = torch.full((n, 3), float(i)).pin_memory()
        b['name'] = f'batch{i}'
        raws.append(b)
    stats = (2841.94, 1417.26)
    stager = BatchStager(raws, DEV, depth_stats=stats)
    busy = torch.randn(4096, 4096, device=DEV)
    seen = []
    for i, b in enumerate(stager):
        for _ in range(4):
            busy = busy @ busy * 1e-3            # the "step": the next copy overlaps this
        seen.append(b['name'])
        assert torch.equal(b['rgb'], normalize_rgb(raws[i]['rgb'].to(DEV)))
        assert torch.equal(b['depth'], normalize_depth(raws[i]['depth'].to(DEV), *stats))
        assert torch.equal(b['label'].cpu(), raws[i]['label'])
        assert b['rgb'].shape == (n, 3, h, w) and b['depth'].shape == (n, 1, h, w)
    assert seen == [f'batch{i}' for i in range(7)]
    assert stager.batches_staged == 7
    assert stager.bytes_staged == 7 * (n * h * w * (3 + 2) + n * 3 * 4)


@pytest.mark.gpu
def test_stager_rejects_pageable_memory():
    from emsanet_amd import _lib
    from emsanet_amd.staging import BatchStager
    b = {'rgb': torch.zeros(1, 32, 32, 3, dtype=torch.uint8)}
    with pytest.raises(_lib.EmsaError):
        next(iter(BatchStager([b], DEV)))


@pytest.mark.gpu
def test_staged_batch_feeds_the_model():
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.staging import BatchStager, pinned_raw_batch
    args = full_args(input_height=64, input_width=96)
    torch.manual_seed(0)
    model = EMSANet(args, nyuv2_config()).to(DEV).eval()
    raws = [pinned_raw_batch(2, 64, 96, seed=i) for i in range(3)]
    outs = []
    with torch.no_grad():
        for b in BatchStager(raws, DEV, depth_stats=(2841.94, 1417.26)):
            outs.append(model(b)[0][0].float().cpu())
    assert all(torch.isfinite(o).all() for o in outs)
    assert not torch.equal(outs[0], outs[1])
