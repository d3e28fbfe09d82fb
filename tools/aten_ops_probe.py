"""which Python lines of the engine still issue ATen kernels (fill / add / copy / cat ...) inside one fp32
training step -- the `at::native::*` rows of the kernel tables"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
from emsanet_amd import full_args, nyuv2_config   # noqa: E402
from emsanet_amd.model import EMSANet   # noqa: E402
from emsanet_amd.optim import FusedSGD   # noqa: E402
from emsanet_amd.parallel import GradientBuckets   # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else 'f32'
dev = torch.device('cuda', 0)
a = full_args(input_height=480, input_width=640,
              compute_dtype={'f32': 'float32', 'bf16': 'bfloat16'}[dt])
torch.manual_seed(0)
model = EMSANet(a, nyuv2_config())
bench.deterministic_init_(model)
model.to(dev).train()
batch = bench.synthetic_batch_device(8, 480, 640, 1234, dev)
params = [p for p in model.parameters() if p.requires_grad]
buckets = GradientBuckets(params)
opt = FusedSGD(buckets, lr=1e-5, momentum=0.9, weight_decay=1e-4)
cots = None


def step():
    global cots
    buckets.reset()
    flat = bench.flatten_outputs(model(batch))
    if cots is None:
        g = torch.Generator(device='cpu').manual_seed(4321)
        cots = [(torch.randn(t.shape, generator=g) * 1e-3).to(dev).contiguous(
            memory_format=torch.channels_last if t.dim() == 4 else torch.contiguous_format) for t in flat]
    torch.autograd.backward(flat, cots)
    buckets.finish()
    opt.step()


import traceback   # noqa: E402

from torch.utils._python_dispatch import TorchDispatchMode   # noqa: E402

WATCH = ('fill_', 'zero_', 'copy_', 'cat', 'add', 'add_', 'zeros', 'zeros_like', 'clone', 'mul', 'sum', 'stack',
         '_foreach_copy_', '_foreach_zero_', 'index', 'slice_backward', 'select_backward', 'new_zeros', 'full')
sites = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        out = func(*args, **(kwargs or {}))
        if name in WATCH:
            t = next((a for a in list(args) + [out] if torch.is_tensor(a)), None)
            if t is not None and t.is_cuda:
                fr = [f for f in traceback.extract_stack() if 'emsanet_amd' in f.filename]
                where = f'{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}' if fr else 'autograd engine (no engine frame)'
                n = t.numel() if torch.is_tensor(t) else 0
                sites[(name, where, 'big' if n > 65536 else 'small')] += 1
        return out


for _ in range(3):
    step()
torch.cuda.synchronize()
with Log():
    step()
torch.cuda.synchronize()
for (name, where, size), n in sites.most_common(60):
    print(f'{n:4d}  {name:16s} {size:5s} {where}')
print('total watched ATen calls on CUDA tensors in one step:', sum(sites.values()))
