"""which gradients are furthest from the fp64 oracle with the weight-free decoder up-sampling modes
(probe behind tests/test_model_gpu.py::test_pinned_gradients_weight_free_decoder_upsampling)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch.nn.functional as F   # noqa: E402

import test_model_gpu as T   # noqa: E402
from emsanet_amd import full_args, nyuv2_config, ops   # noqa: E402
from emsanet_amd.model import EMSANet   # noqa: E402
from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict, synthetic_batch   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'bilinear'
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 23
args = full_args(input_height=96, input_width=128, tasks=('semantic', 'instance', 'orientation', 'scene', 'normal'),
                 semantic_decoder_upsampling=mode, instance_decoder_upsampling=mode,
                 normal_decoder_upsampling=mode, upsampling_prediction=mode)
cfg = nyuv2_config()
oracle = EMSANetOracle(args, cfg)
sd = deterministic_state_dict(oracle, 0)
oracle.load_state_dict(sd)
oracle = oracle.double()
model = EMSANet(args, cfg)
model.load_state_dict(sd)
model.to('cuda:0')
for m in (model, oracle):
    m.train()
    m.dropout_seed = seed
batch = synthetic_batch(3, 96, 128)
ops.MASK_TRACE = []
out = model({k: v.to('cuda:0') for k, v in batch.items()})
trace, ops.MASK_TRACE = ops.MASK_TRACE, None
pinned = T._PinnedRelu(trace)
orig = F.relu
F.relu = pinned
ref = oracle({k: v.double() for k, v in batch.items()})
F.relu = orig
fo, fr = T._flatten(out), T._flatten(ref)
cots = [T.rnd(*t.shape, seed=100 + i, scale=1e-1) for i, t in enumerate(fr)]
torch.autograd.backward(fo, [c.to('cuda:0') for c in cots])
torch.autograd.backward(fr, [c.double() for c in cots])
pr = dict(oracle.named_parameters())
gmax = max(p.grad.abs().max().item() for p in pr.values() if p.grad is not None)
rows = []
for k, p in model.named_parameters():
    r = pr[k].grad.double()
    g = p.grad.detach().cpu().double()
    if r.abs().max().item() < 1e-9 * gmax or k.endswith(('conv1x3_1.bias', 'conv1x3_2.bias')):
        continue
    rows.append(((g - r).norm().item() / r.norm().item(), k, r.norm().item(), r.numel()))
rows.sort(reverse=True)
print(f"mode {mode} seed {seed}: gmax {gmax:.3e}")
for e, k, nrm, n in rows[:12]:
    print(f"  {e:.3e}  |ref| {nrm:.3e}  n {n:6d}  {k}")
