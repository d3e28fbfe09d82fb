"""Debug aid (GPU box): where does the first replay of GraphedTrainStep leave the eager twin?
usage: python tools/graph_step_debug.py <experiment>
  keep      warm-up updates kept (round-2 behaviour), twin takes the same 3 eager steps
  restore   default: warm-up effects taken back
  no_buf    restore without the BatchNorm buffers      no_drop  restore without the Dropout2d step
  no_par    restore without parameters / momentum"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsanet_amd import full_args, nyuv2_config, graph as G          # noqa: E402
from emsanet_amd.model import EMSANet                                 # noqa: E402
from emsanet_amd.optim import FusedSGD                                # noqa: E402
from emsanet_amd.parallel import GradientBuckets                     # noqa: E402
from oracle.emsanet_oracle import synthetic_batch                     # noqa: E402

DEV = 'cuda:0'
exp = sys.argv[1] if len(sys.argv) > 1 else 'restore'
args = full_args(input_height=96, input_width=128)


def flat(out):
    r = []
    for o, sides in out:
        r += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            r += list(s) if isinstance(s, tuple) else [s]
    return r


def build():
    torch.manual_seed(0)
    m = EMSANet(args, nyuv2_config()).to(DEV).train()
    m.dropout_seed = 99
    b = GradientBuckets([p for p in m.parameters() if p.requires_grad])
    o = FusedSGD(b, lr=0.0, momentum=0.9, weight_decay=0.0)
    return m, b, o


def loss_of(out):
    return sum((t * t).mean() for t in flat(out))


taps = {}


HOOKS = os.environ.get('HOOKS', 'prealloc')     # none | clone | prealloc


def hook_all(model, store):
    """clone: fresh tensors per call (inside a capture they come from the graph's pool and change
    its allocation pattern); prealloc: buffers allocated by the first (eager) call, later calls only
    copy into them -- the capture allocates nothing extra"""
    def mk(name):
        def h(mod, inp, out):
            ts = [out] if torch.is_tensor(out) else [t for t in (out if isinstance(out, (list, tuple)) else []) if torch.is_tensor(t)]
            if not ts:
                return
            if HOOKS == 'clone' or name not in store:
                store[name] = [t.detach().clone() for t in ts]
            else:
                for b, t in zip(store[name], ts):
                    b.copy_(t.detach())
        return h
    if HOOKS == 'none':
        return
    depth = int(os.environ.get('HOOK_DEPTH', '2'))
    for name, mod in model.named_modules():
        if name and name.count('.') <= depth:
            mod.register_forward_hook(mk(name))


batches = [{k: v.to(DEV) for k, v in synthetic_batch(4, 96, 128, seed=s).items()} for s in (1, 2)]
m2, b2, o2 = build()
m3, b3, o3 = build()
s2, s3 = {}, {}
hook_all(m2, s2)
hook_all(m3, s3)


def eager_step(m, b, o, batch):
    b.reset()
    out = m(batch)
    loss = loss_of(out)
    loss.backward()
    b.finish()
    o.step()
    return float(loss.detach()), [t.detach().clone() for t in flat(out)]


if exp != 'restore' and exp != 'keep':
    orig = G._TrainStateSnapshot.restore

    @torch.no_grad()
    def restore(self):
        if exp == 'no_buf':
            self.buffers = []
        if exp == 'no_par':
            self.params, self.momentum = [], []
        if exp == 'no_drop':
            self.dropout_step = self.model.dropout_step
        orig(self)
    G._TrainStateSnapshot.restore = restore

PRE = os.environ.get('PRE_SD') == '1'       # test-like: state_dict clone before the capture
POST = os.environ.get('POST_SD') == '1'     # test-like: state_dict compare after the capture
if PRE:
    sd0 = {k: v.clone() for k, v in m2.state_dict().items()}
if exp == 'keep':
    for _ in range(3):
        eager_step(m3, b3, o3, batches[0])
g = G.GraphedTrainStep(m2, batches[0], b2, o2, loss_fn=loss_of, warmup=3,
                       keep_warmup_updates=exp == 'keep')
if exp == 'no_drop':
    m3.dropout_step = m2.dropout_step
torch.cuda.synchronize()
print('graph:', g.graph_info)
if POST:
    sd1 = m2.state_dict()
    if PRE:
        bad = [k for k, v in sd1.items() if not torch.equal(v, sd0[k])]
        print('state entries changed by building the graph:', bad[:8])
print(exp, 'PRE', PRE, 'POST', POST, 'dropout steps', m2.dropout_step, m3.dropout_step, 'first', o2._first, o3._first)
l2, out2 = g.replay(batches[1])
torch.cuda.synchronize()
print(exp, 'static loss right after the replay', float(l2.detach()),
      '| eager loss_of(static outputs)', float(loss_of(out2).detach()))
parts_static = [float((t * t).mean()) for t in flat(out2)]
out2 = [t.clone() for t in flat(out2)]
l3, out3 = eager_step(m3, b3, o3, batches[1])
torch.cuda.synchronize()
print(exp, 'loss graph', float(l2), 'eager', l3)
print(exp, 'per-output (t*t).mean() of the static outputs:', [round(v, 4) for v in parts_static])
l2b, _ = g.replay(batches[1])
torch.cuda.synchronize()
print(exp, 'second replay of the same batch: static loss', float(l2b.detach()))
for i, (a, b) in enumerate(zip(out2, out3)):
    print('  output', i, tuple(a.shape), 'max diff', float((a - b).abs().max()), 'max', float(b.abs().max()))
first = None
for name in s3:
    if name in s2:
        d = max(float((a - b).abs().max()) for a, b in zip(s2[name], s3[name]))
        if d > 0 and first is None:
            first = name
        if d > 0:
            print('  module', name, 'diff', d)
print(exp, 'first diverging module:', first)
