"""Run-to-run reproducibility of the eval-mode (frozen BatchNorm) backward pass at bs 32, 640x480:
same inputs, same cotangents, twice -- per-tensor rel-L2 between the two runs.
usage: python tools/grad_repeat_probe.py [f32|bf16] [bs]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from emsanet_amd import _lib, full_args                          # noqa: E402
import test_timed_size_gpu as T                                  # noqa: E402


def main():
    dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == 'bf16') else torch.float32
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    _lib.lib().emsa_set_batch_invariant(1)
    model = T._model(full_args(), None if dtype == torch.float32 else dtype)
    batch = T._inputs(bs, 480, 640, seed=11)
    T._recalibrate(model, batch)
    with torch.no_grad():
        shapes = [t.shape for t in T._flatten(model(batch))]
    g = torch.Generator().manual_seed(4321)
    cots = [(torch.randn(s, generator=g) * 1e-1).to('cuda:0') for s in shapes]
    from emsanet_amd import ops
    trace = os.environ.get('TRACE') == '1'
    runs, traces = [], []
    for _ in range(3 if not trace else 2):
        if trace:
            ops.TRACE = []
        runs.append(T._grads(model, batch, cots))
        if trace:
            traces.append(ops.TRACE)
            ops.TRACE = None
    if trace:
        shown = 0
        for i, ((ka, a), (kb, b)) in enumerate(zip(traces[1], traces[0])):
            assert ka == kb and a.shape == b.shape
            if not torch.equal(a, b) and (a.dtype != torch.float32 or a.dim() == 4 and a.shape[0] == bs):
                af, bf = a.float(), b.float()
                d = (af - bf).abs().max().item() / max(1e-30, bf.abs().max().item())
                nd = int((af != bf).sum())
                print(f"  trace #{i} {ka} shape {tuple(a.shape)} {a.dtype} max rel diff {d:.3e}, {nd} of {a.numel()} elements differ")
                shown += 1
                if shown >= 12:
                    break
        print(f"{len(traces[0])} traced gradient tensors")
    for j in range(1, len(runs)):
        rows = []
        for k, a in runs[0].items():
            b = runs[j][k]
            if float(b.norm()) == 0:
                continue
            rows.append((float((a - b).norm() / b.norm()), k))
        rows.sort(reverse=True)
        print(f"run 0 vs run {j} ({dtype}, bs {bs}, EMSA_DUAL_STREAM={os.environ.get('EMSA_DUAL_STREAM')}): "
              f"{sum(1 for r in rows if r[0] > 0)} of {len(rows)} tensors differ; worst:")
        for r in rows[:5]:
            print("   %.3e  %s" % r)


if __name__ == '__main__':
    main()
