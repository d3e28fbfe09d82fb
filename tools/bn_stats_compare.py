"""Debug tool (GPU box): batch statistics of every BatchNorm and the scale of every module output,
engine (bf16 storage) vs the fp64 oracle (storage-emulating / plain), one train-mode forward on the
engine's ReLU branch.  BatchNorm momentum is set to 1 on both sides, so after the forward
`running_mean` / `running_var` ARE the batch statistics.  A BatchNorm hides a scale error of its
input in the forward pass (the output is renormalised) and returns it, inverted, in the backward
pass (dx ~ invstd): this is the forward-side view of tools/actgrad_compare.py.

  python tools/bn_stats_compare.py [H W BS] [--plain] [--out FILE]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from emsanet_amd import full_args, nyuv2_config, ops            # noqa: E402
from emsanet_amd.model import EMSANet                            # noqa: E402
from oracle import emsanet_oracle as O                           # noqa: E402
from test_model_gpu import _PinnedRelu                           # noqa: E402

DEV = 'cuda:0'


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    h, w, bs = (int(argv[0]), int(argv[1]), int(argv[2])) if len(argv) >= 3 else (256, 320, 8)
    plain = '--plain' in sys.argv
    f32 = '--f32' in sys.argv
    out_path = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else None
    args = full_args(input_height=h, input_width=w)
    cfg = nyuv2_config()
    oracle = O.EMSANetOracle(args, cfg)
    oracle.load_state_dict(O.deterministic_state_dict(oracle, 0))
    sd = oracle.state_dict()
    oracle = oracle.double()
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    model.to(DEV)
    if not f32:
        model.set_compute_dtype(torch.bfloat16)
    for m in (model, oracle):
        m.train()
        m.dropout_seed = 321
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.momentum = 1.0
    batch = O.synthetic_batch(bs, h, w)
    fe, fo = {}, {}

    def hooks(net, store, conv):
        hs = []
        for name, mod in net.named_modules():
            if name:
                hs.append(mod.register_forward_hook(
                    lambda _m, _i, out, name=name: store.__setitem__(name, conv(out))
                    if torch.is_tensor(out) else None))
        return hs
    hs = hooks(model, fe, lambda t: t.detach().float().cpu().double()) + \
        hooks(oracle, fo, lambda t: t.detach().clone())
    ops.MASK_TRACE = []
    model({k: v.to(DEV) for k, v in batch.items()})
    trace = ops.MASK_TRACE
    ops.MASK_TRACE = None
    pinned = _PinnedRelu(trace)
    relu0 = F.relu
    F.relu = pinned
    if not plain and not f32:
        O.Spec.STORAGE = torch.bfloat16
    try:
        oracle({k: v.double() for k, v in batch.items()})
    finally:
        F.relu = relu0
        O.Spec.STORAGE = None
    torch.cuda.synchronize()
    for hnd in hs:
        hnd.remove()
    lines = [f"# bn_stats_compare {'f32' if f32 else 'bf16'}{' plain-oracle' if plain else ''} {h}x{w} bs {bs}",
             "# BatchNorm batch statistics (forward order): var engine / var oracle, "
             "|mean_e - mean_o| / std_o (max over channels), name"]
    eb = dict(model.named_buffers())
    for k, b in oracle.named_buffers():
        if k.endswith('running_var'):
            ve, vo = eb[k].detach().cpu().double(), b.double()
            me = eb[k.replace('running_var', 'running_mean')].detach().cpu().double()
            mo = dict(oracle.named_buffers())[k.replace('running_var', 'running_mean')].double()
            r = ve / vo
            lines.append("%.4f %.4f %.4f  %.2e  bn %s" % (r.median().item(), r.min().item(), r.max().item(),
                                                          ((me - mo).abs() / vo.sqrt()).max().item(),
                                                          k[:-len('.running_var')]))
    lines.append("# module outputs (forward order): norm engine / norm oracle, rel-L2 error, name")
    for n, _ in oracle.named_modules():
        if n in fe and n in fo and fe[n].shape == fo[n].shape:
            a, b = fe[n], fo[n].cpu()
            lines.append("%.4f %.2e out %s" % (a.norm().item() / max(1e-300, b.norm().item()),
                                               (a - b).norm().item() / max(1e-300, b.norm().item()), n))
    text = '\n'.join(lines)
    print(text)
    if out_path:
        with open(out_path, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
