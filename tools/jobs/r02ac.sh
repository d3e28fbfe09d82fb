#!/bin/bash
# debugging the reverted SE vectorisation: poisoned eager step with the variant library
O=gpurun_out/r02ac; mkdir -p $O
cat > /tmp/poison_step.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from emsanet_amd import full_args, nyuv2_config
from emsanet_amd.model import EMSANet
from oracle.emsanet_oracle import synthetic_batch
args = full_args(input_height=96, input_width=128)
torch.manual_seed(0)
m = EMSANet(args, nyuv2_config()).to('cuda:0').train()
m.dropout_seed = 99
batch = {k: v.to('cuda:0') for k, v in synthetic_batch(4, 96, 128, seed=1).items()}
def flat(out):
    r = []
    for o, sides in out:
        r += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            r += list(s) if isinstance(s, tuple) else [s]
    return r
for it in range(2):
    out = flat(m(batch))
    loss = sum((t * t).mean() for t in out)
    loss.backward()
    bad = [k for k, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    print('iter', it, 'loss', float(loss), 'non-finite grads:', len(bad), bad[:6])
    for p in m.parameters():
        p.grad = None
PY
echo "== default lib, poison"; EMSA_POISON=1 timeout 300 python /tmp/poison_step.py 2>&1 | tail -3
echo "== SE variant lib, poison"; EMSA_POISON=1 EMSA_LIB=$GRAFT_REPO_ROOT/emsanet_amd/lib/var_se/libemsanet_hip.so timeout 300 python /tmp/poison_step.py 2>&1 | tail -3
echo "== SE variant lib, no poison"; EMSA_LIB=$GRAFT_REPO_ROOT/emsanet_amd/lib/var_se/libemsanet_hip.so timeout 300 python /tmp/poison_step.py 2>&1 | tail -3
