"""GPU diagnostic: the same engine in fp32 and in a 16-bit storage type on the same batch; relative
L2 difference of every block output in execution order (shows where 16-bit storage noise grows).
usage: python tools/stagewise_dtype.py [bf16|f16] [train|eval] [bs h w]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.nn import ConvNormAct, NonBottleneck1D, SEAddUniRGB
    from emsanet_amd.decoder import DecoderModule
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import deterministic_state_dict
    dt = {'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
    train = (sys.argv[2] if len(sys.argv) > 2 else 'train') == 'train'
    bs, h, w = [int(v) for v in sys.argv[3:6]] if len(sys.argv) > 5 else (2, 480, 640)
    args = full_args(input_height=h, input_width=w)
    models, recs = [], []
    sd = None
    for d in (torch.float32, dt):
        torch.manual_seed(0)
        m = EMSANet(args, nyuv2_config())
        if sd is None:
            sd = deterministic_state_dict(m, 0)
        m.load_state_dict(sd)
        m.to('cuda:0').set_compute_dtype(d)
        m.train(train)
        m.dropout_seed = 5
        rec = []
        for name, mod in m.named_modules():
            if isinstance(mod, (NonBottleneck1D, ConvNormAct, SEAddUniRGB, DecoderModule)):
                mod.register_forward_hook(
                    lambda mod_, inp, out, name=name, rec=rec: rec.append(
                        (name, (out[0] if isinstance(out, tuple) else out).detach().float().clone())))
        models.append(m)
        recs.append(rec)
    import numpy as np
    rng = np.random.default_rng(1234)
    rgb = rng.integers(0, 255, (bs, h, w, 3), dtype=np.uint8).astype(np.float32) / 255
    depth = rng.integers(0, 40000, (bs, h, w), dtype=np.uint16).astype(np.float32) / 20000
    batch = {'rgb': torch.from_numpy(rgb.transpose(0, 3, 1, 2).copy()).cuda(),
             'depth': torch.from_numpy(depth[:, None].copy()).cuda()}
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        for m in models:
            m(batch)
    for (n0, a), (n1, b) in zip(*recs):
        assert n0 == n1
        e = (a - b).norm().item() / max(1e-30, a.norm().item())
        print(f"{e:9.3e}  {n0}  |x|={a.norm().item() / a.numel() ** 0.5:.3e}")


if __name__ == '__main__':
    main()
