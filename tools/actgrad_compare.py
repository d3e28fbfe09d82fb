"""Debug tool (GPU box): where do the ACTIVATION gradients of the engine and of the oracle part?

VERDICT r4 weak 2: in bf16 training the engine's parameter-gradient norms sit 1.00 (heads) ...
1.09 (every encoder tensor) above the storage-emulating fp64 oracle's, at per-tensor cosine 0.996
-- too systematic for uncorrelated rounding noise.  Parameter gradients only show where a tensor's
gradient ends up; this tool compares the gradients that FLOW:

  * per convolution module: the gradient w.r.t. the conv's output as its backward kernels receive
    it (engine: ops.GRAD_TRACE, written in _conv_backward / StemFunction / MultiConvFunction;
    oracle: a tensor hook on every conv output, by wrapping oracle.conv_q / nn.Conv2d.forward);
  * per module boundary: the gradient w.r.t. the output of every module both models have under the
    same name (NBt1D blocks, decoder modules, up-sampling, skip fusion, SE fusion, context module).

Printed in backward order (the order the gradient travels): norm ratio engine / oracle, cosine,
and the same two numbers for the per-channel pixel sums (what a bias gradient sees: rounding noise
averages out of it, a systematic scale does not).

  python tools/actgrad_compare.py [bf16|f32] [H W BS] [--plain] [--eval] [--out FILE]
    bf16 (default): engine in bf16 storage vs the fp64 oracle in storage-emulation mode, both on
                    the engine's ReLU branch; --plain: against the oracle WITHOUT storage emulation
    f32:            control -- fp32 engine vs fp64 oracle (ratios must be 1.000)
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
from util import rnd                                             # noqa: E402

DEV = 'cuda:0'


def flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def stats(e, o):
    """norm ratio, cosine, and both again for the per-channel sums over (N, H, W)"""
    e, o = e.double(), o.double()
    if e.shape != o.shape:
        return None
    ne, no = e.norm().item(), o.norm().item()
    cos = float((e * o).sum() / max(1e-300, ne * no))
    if e.dim() == 4:
        se, so = e.sum((0, 2, 3)), o.sum((0, 2, 3))
        rs = se.norm().item() / max(1e-300, so.norm().item())
        cs = float((se * so).sum() / max(1e-300, se.norm().item() * so.norm().item()))
    else:
        rs, cs = float('nan'), float('nan')
    return ne / max(1e-300, no), cos, rs, cs


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    mode = argv[0] if argv else 'bf16'
    h, w, bs = (int(argv[1]), int(argv[2]), int(argv[3])) if len(argv) >= 4 else (256, 320, 8)
    plain = '--plain' in sys.argv
    out_path = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else None
    args = full_args(input_height=h, input_width=w)
    cfg = nyuv2_config()
    oracle = O.EMSANetOracle(args, cfg)
    sd = O.deterministic_state_dict(oracle, 0)
    oracle.load_state_dict(sd)
    oracle = oracle.double()
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    model.to(DEV)
    if mode == 'bf16':
        model.set_compute_dtype(torch.bfloat16)
    for m in (model, oracle):
        m.train('--eval' not in sys.argv)      # --eval: frozen BatchNorm statistics, gradients on
        m.dropout_seed = 321
    batch = O.synthetic_batch(bs, h, w)

    # ---- module-boundary hooks, both sides ---------------------------------------------------
    def boundary_hooks(net, store, to_cpu):
        hs = []
        for name, mod in net.named_modules():
            if not name:
                continue

            def fwd_hook(_m, _i, out, name=name):
                if torch.is_tensor(out) and out.requires_grad:
                    out.register_hook(lambda g, name=name: store.__setitem__(name, to_cpu(g)))
            hs.append(mod.register_forward_hook(fwd_hook))
        return hs
    be, bo = {}, {}
    order = []
    he = boundary_hooks(model, be, lambda g: g.detach().float().cpu())
    ho = boundary_hooks(oracle, bo, lambda g: g.detach().clone())

    # ---- engine forward (records its ReLU decisions) -----------------------------------------
    ops.MASK_TRACE = []
    ops.GRAD_TRACE = {}
    out = flatten(model({k: v.to(DEV) for k, v in batch.items()}))
    trace = ops.MASK_TRACE
    ops.MASK_TRACE = None

    # ---- oracle forward on the same branch, conv outputs hooked --------------------------------
    conv_name = {id(m): n for n, m in oracle.named_modules() if isinstance(m, torch.nn.Conv2d)}
    eng_conv = {n: id(m) for n, m in model.named_modules() if isinstance(m, torch.nn.Conv2d)}
    co = {}
    conv_order = []
    orig_conv_q, orig_fwd = O.conv_q, torch.nn.Conv2d.forward

    def hook_out(conv, y):
        n = conv_name.get(id(conv))
        if n is not None and y.requires_grad:
            conv_order.append(n)
            y.register_hook(lambda g, n=n: co.__setitem__(n, g.detach().clone()))
        return y
    O.conv_q = lambda conv, x: hook_out(conv, orig_conv_q(conv, x))
    torch.nn.Conv2d.forward = lambda self, x: hook_out(self, orig_fwd(self, x))
    pinned = _PinnedRelu(trace)
    relu0 = F.relu
    F.relu = pinned
    if mode == 'bf16' and not plain:
        O.Spec.STORAGE = torch.bfloat16
    try:
        ref = flatten(oracle({k: v.double() for k, v in batch.items()}))
    finally:
        torch.nn.Conv2d.forward = orig_fwd
        O.conv_q = orig_conv_q
    assert pinned.i == len(trace), (pinned.i, len(trace))
    cots = [rnd(*t.shape, seed=100 + i, scale=1e-1) for i, t in enumerate(ref)]
    torch.autograd.backward(out, [c.to(DEV) for c in cots])
    torch.autograd.backward(ref, [c.double() for c in cots])
    F.relu = relu0
    O.Spec.STORAGE = None
    torch.cuda.synchronize()
    for hnd in he + ho:
        hnd.remove()
    gt = ops.GRAD_TRACE
    ops.GRAD_TRACE = None

    lines = [f"# actgrad_compare {mode}{' plain-oracle' if plain else ''} {h}x{w} bs {bs}: engine / oracle, "
             f"{pinned.flips} of {pinned.total} ReLU decisions differ from the oracle's own",
             "# columns: norm ratio, cosine, ratio of per-channel pixel sums, cosine of those sums, name",
             "# --- gradient w.r.t. conv OUTPUTS, in backward order ---"]
    seen = set()
    for n in reversed(conv_order):
        if n in seen or n not in co or eng_conv.get(n) not in gt:
            continue
        seen.add(n)
        st = stats(gt[eng_conv[n]], co[n].cpu())
        if st:
            lines.append("%.4f %.4f %.4f %.4f conv %s" % (*st, n))
    lines.append("# --- gradient w.r.t. module OUTPUTS (same-named modules), in backward order ---")
    names = [n for n, _ in oracle.named_modules() if n in be and n in bo]
    for n in reversed(names):
        st = stats(be[n], bo[n].cpu())
        if st:
            lines.append("%.4f %.4f %.4f %.4f module %s" % (*st, n))
    # parameter gradients, for reference
    lines.append("# --- parameter gradients ---")
    pr = dict(oracle.named_parameters())
    for k, p in model.named_parameters():
        if p.grad is None or pr[k].grad is None:
            continue
        g, r = p.grad.detach().cpu().double().flatten(), pr[k].grad.double().flatten()
        if r.norm() < 1e-30:
            continue
        lines.append("%.4f %.4f param %s" % (g.norm().item() / r.norm().item(),
                                             float(torch.dot(g, r) / (g.norm() * r.norm())), k))
    text = '\n'.join(lines)
    print(text)
    if out_path:
        os.makedirs(os.path.dirname(out_path) or '.', exist_ok=True)
        with open(out_path, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
