"""does a ResNet-50 bottleneck model run through the engine (which kernel refuses 1024 / 2048 channels)?"""
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from emsanet_amd import full_args, nyuv2_config   # noqa: E402
from emsanet_amd.model import EMSANet   # noqa: E402
from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict, synthetic_batch   # noqa: E402

dt = {'f32': torch.float32, 'bf16': torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else 'f32']
args = full_args(input_height=96, input_width=128, rgb_encoder_backbone='resnet50',
                 depth_encoder_backbone='resnet50', rgb_encoder_backbone_resnet_block='bottleneck',
                 depth_encoder_backbone_resnet_block='bottleneck')
cfg = nyuv2_config()
oracle = EMSANetOracle(args, cfg)
sd = deterministic_state_dict(oracle, 0)
oracle.load_state_dict(sd)
model = EMSANet(args, cfg)
model.load_state_dict(sd)
model.to('cuda:0')
model.set_compute_dtype(dt)
batch = synthetic_batch(2, 96, 128)
dev = {k: v.to('cuda:0') for k, v in batch.items()}


def flat(o):
    r = []
    for x in o:
        if torch.is_tensor(x):
            r.append(x)
        elif isinstance(x, (tuple, list)):
            r += flat(x)
    return r


for mode in ('eval', 'train'):
    try:
        model.train(mode == 'train'); oracle.train(mode == 'train')
        model.dropout_seed = oracle.dropout_seed = 5
        model.dropout_step = oracle.dropout_step = 0
        with torch.set_grad_enabled(mode == 'train'):
            out = flat(model(dev))
            ref = flat(oracle(batch))
        errs = [float((a.float().cpu() - b).norm() / b.norm()) for a, b in zip(out, ref)]
        print(mode, 'forward rel-L2', ' '.join(f'{e:.1e}' for e in errs))
        if mode == 'train':
            sum((t * t).mean() for t in out).backward()
            sum((t * t).mean() for t in ref).backward()
            pr = dict(oracle.named_parameters())
            rows = []
            for k, p in model.named_parameters():
                r = pr[k].grad
                if r is None or float(r.norm()) == 0:
                    continue
                rows.append((float((p.grad.cpu() - r).norm() / r.norm()), k))
            rows.sort(reverse=True)
            print('train backward worst:', rows[:5])
    except Exception:
        traceback.print_exc()
