"""Which module first makes a sample's eval forward depend on the batch it sits in?
bs-32 vs bs-8 forward of configs[1] with emsa_set_batch_invariant(1); forward hooks keep the first
8 samples of every module's output in the bs-32 pass and compare bit for bit in the bs-8 pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from emsanet_amd import _lib, full_args, nyuv2_config          # noqa: E402
from emsanet_amd.model import EMSANet                           # noqa: E402
from util import deterministic_state_dict                       # noqa: E402


def first(o):
    if torch.is_tensor(o):
        return o
    if isinstance(o, (list, tuple)):
        for v in o:
            t = first(v)
            if t is not None:
                return t
    if isinstance(o, dict):
        for v in o.values():
            t = first(v)
            if t is not None:
                return t
    return None


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
    _lib.lib().emsa_set_batch_invariant(int(os.environ.get('INVARIANT', '1')))
    dev = 'cuda:0'
    model = EMSANet(full_args(), nyuv2_config())
    model.load_state_dict(deterministic_state_dict(model))
    model.to(dev).eval()
    if dtype != 'f32':
        model.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(11)
    batch = {'rgb': torch.randn(32, 3, 480, 640, generator=g).to(dev),
             'depth': torch.randn(32, 1, 480, 640, generator=g).to(dev)}
    kept, order, mode = {}, [], ['keep']
    report = []

    def hook(name):
        def fn(mod, inp, out):
            t = first(out)
            if t is None or t.dim() < 2 or t.shape[0] not in (8, 32):
                return
            if mode[0] == 'keep':
                kept[name] = t[:8].detach().clone()
                order.append(name)
            elif name in kept:
                a, b = t.detach(), kept[name]
                if a.shape == b.shape and not torch.equal(a, b):
                    d = (a.float() - b.float()).abs().max().item()
                    report.append((name, type(mod).__name__, tuple(a.shape), d,
                                   b.float().abs().max().item()))
        return fn
    for name, mod in model.named_modules():
        if name:
            mod.register_forward_hook(hook(name))
    with torch.no_grad():
        model(batch)
        mode[0] = 'cmp'
        model({k: v[:8].contiguous() for k, v in batch.items()})
    print(f"{len(order)} module outputs compared, {len(report)} differ")
    idx = {n: i for i, n in enumerate(order)}
    for name, typ, shape, d, mx in sorted(report, key=lambda r: idx[r[0]])[:25]:
        print(f"  #{idx[name]:4d} {name:70s} {typ:28s} {shape} max|diff| {d:.3e} of {mx:.3e}")




def probe_final_conv():
    _lib.lib().emsa_set_batch_invariant(1)
    dev = 'cuda:0'
    model = EMSANet(full_args(), nyuv2_config())
    model.load_state_dict(deterministic_state_dict(model))
    model.to(dev).eval()
    g = torch.Generator().manual_seed(3)
    from emsanet_amd import functional as Fn
    cat = Fn.as_act(torch.randn(32, 1024, 15, 20, generator=g).to(dev))
    x512 = Fn.as_act(torch.randn(32, 512, 15, 20, generator=g).to(dev))
    with torch.no_grad():
        for name, mod, t in (('final_conv', model.context_module.final_conv, cat),
                             ('ppm feat conv 5x5', model.context_module.features[1][1],
                              Fn.as_act(torch.randn(32, 512, 5, 5, generator=g).to(dev))),
                             ('dec conv3x3', model.decoders['semantic_decoder'].decoder_modules[0].conv3x3, x512)):
            a = mod(t)
            for nb in (8, 16, 24):
                b = mod(Fn.as_act(t[:nb].contiguous()))
                print(name, nb, 'equal' if torch.equal(a[:nb], b) else
                      'DIFF %.3e' % (a[:nb] - b).abs().max().item())


if len(sys.argv) > 2 and sys.argv[2] == 'conv':
    probe_final_conv()

if __name__ == '__main__' and len(sys.argv) <= 2:
    main()


def probe_ppm():
    _lib.lib().emsa_set_batch_invariant(1)
    dev = 'cuda:0'
    from emsanet_amd import functional as Fn, ops
    model = EMSANet(full_args(), nyuv2_config())
    model.load_state_dict(deterministic_state_dict(model))
    model.to(dev).eval()
    g = torch.Generator().manual_seed(3)
    x = Fn.as_act(torch.randn(32, 512, 15, 20, generator=g).to(dev))
    xs = Fn.as_act(x[:8].contiguous())
    cm = model.context_module
    eq = lambda a, b: 'equal' if torch.equal(a[:8], b) else 'DIFF %.3e' % (a[:8].float() - b.float()).abs().max().item()   # noqa: E731
    with torch.no_grad():
        fa, fb = [], []
        for b, f in zip(cm.bins, cm.features):
            pa, pb = ops.AdaptiveAvgPoolFunction.apply(x, b), ops.AdaptiveAvgPoolFunction.apply(xs, b)
            print('pool', b, eq(pa, pb))
            ca, cb = f[1](pa), f[1](pb)
            print('feat conv', b, eq(ca, cb), tuple(ca.shape))
            cb2 = f[1](Fn.as_act(pa[:8].contiguous()))
            print('feat conv on identical input', b, eq(ca, cb2))
            fa.append(ca)
            fb.append(cb)
        cata, catb = ops.PPMConcatFunction.apply(x, *fa), ops.PPMConcatFunction.apply(xs, *fb)
        print('cat', eq(cata, catb))
        print('final', eq(cm.final_conv(cata), cm.final_conv(catb)))


if len(sys.argv) > 2 and sys.argv[2] == 'ppm':
    probe_ppm()
