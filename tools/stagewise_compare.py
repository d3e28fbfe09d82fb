"""Debug tool (GPU box): run engine and oracle on the same inputs/weights, compare the output of
every equally-named sub-module in execution order and print the relative error per stage.
usage: python tools/stagewise_compare.py [train|eval] [H W] [dropout 0/1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from emsanet_amd import full_args, nyuv2_config            # noqa: E402
from emsanet_amd.model import EMSANet                       # noqa: E402
from oracle.emsanet_oracle import (EMSANetOracle, deterministic_state_dict,   # noqa: E402
                                   synthetic_batch)


def first_tensor(o):
    if torch.is_tensor(o):
        return o
    if isinstance(o, (tuple, list)):
        for x in o:
            t = first_tensor(x)
            if t is not None:
                return t
    return None


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
    h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 160)
    drop = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    kw = {} if drop else dict(dropout_p=0.0, semantic_decoder_block_dropout_p=0.0,
                              instance_decoder_block_dropout_p=0.0)
    args = full_args(input_height=h, input_width=w, **kw)
    cfg = nyuv2_config()
    oracle = EMSANetOracle(args, cfg)
    sd = deterministic_state_dict(oracle, 0)
    oracle.load_state_dict(sd)
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    model.to('cuda:0')
    import copy
    o32 = copy.deepcopy(oracle)
    oracle.double()
    oracle.train(mode == 'train'), model.train(mode == 'train'), o32.train(mode == 'train')
    rec_o, rec_m, rec_c, order = {}, {}, {}, []

    def hook(store, name, keep_order):
        def f(mod, inp, out):
            t = first_tensor(out)
            if t is not None:
                store[name] = t.detach().double().cpu()
                if keep_order:
                    order.append(name)
        return f
    for n, m in oracle.named_modules():
        if n:
            m.register_forward_hook(hook(rec_o, n, True))
    for n, m in model.named_modules():
        if n:
            m.register_forward_hook(hook(rec_m, n, False))
    for n, m in o32.named_modules():
        if n:
            m.register_forward_hook(hook(rec_c, n, False))
    batch = synthetic_batch(2, h, w)
    ctx = torch.no_grad() if mode == 'eval' else torch.enable_grad()
    with ctx:
        oracle({k: v.double() for k, v in batch.items()})
        model({k: v.to('cuda:0') for k, v in batch.items()})
        o32(batch)
    print(f"{'module':75s} rel_err   max|ref|   cpu32_err")
    for n in order:
        if n not in rec_m:
            continue
        a, b = rec_m[n], rec_o[n]
        if a.shape != b.shape:
            a = a[:, :b.shape[1]] if a.dim() == b.dim() and a.shape[1] >= b.shape[1] else a
        if a.shape != b.shape:
            print(f"{n:75s} shape {tuple(a.shape)} vs {tuple(b.shape)}")
            continue
        den = max(b.abs().max().item(), 1e-30)
        err = (a - b).abs().max().item() / den
        flag = '  <<<' if err > 1e-3 else ''
        c = rec_c.get(n)
        cerr = (c - b).abs().max().item() / den if c is not None and c.shape == b.shape else -1
        print(f"{n:75s} {err:.2e}  {den:.3e}  {cerr:.2e}{flag}")


if __name__ == '__main__':
    main()
