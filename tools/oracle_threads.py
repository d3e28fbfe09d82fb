"""how long the CPU oracle takes on this host as a function of torch's thread count (the GPU test
suite's time is the oracle's: VERDICT r4 weak 9)"""
import sys
import time

import torch

sys.path.insert(0, '.')
from emsanet_amd import full_args, nyuv2_config                                   # noqa: E402
from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict, synthetic_batch   # noqa: E402

for (h, w, bs, dt) in ((96, 128, 4, torch.float64), (256, 320, 8, torch.float64), (480, 640, 2, torch.float64),
                       (480, 640, 2, torch.float32)):
    args = full_args(input_height=h, input_width=w)
    o = EMSANetOracle(args, nyuv2_config())
    o.load_state_dict(deterministic_state_dict(o, 0))
    o = o.to(dt).train()
    b = {k: v.to(dt) for k, v in synthetic_batch(bs, h, w).items()}
    for nt in (8, 16, 32, 64, 128):
        if nt > torch.get_num_threads() and nt > 64:
            pass
        torch.set_num_threads(nt)
        t = time.time()
        outs = o(b)
        flat = [x for o_, s in outs for x in (list(o_) if isinstance(o_, tuple) else [o_])]
        sum((x * x).mean() for x in flat).backward()
        print(f"{h}x{w} bs{bs} {str(dt)[6:]} threads {nt}: fwd+bwd {time.time() - t:.2f} s", flush=True)
