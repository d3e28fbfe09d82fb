#!/bin/bash
O=gpurun_out/r04_bn2; mkdir -p $O; R=$PWD
cat > /tmp/bnmicro.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from emsanet_amd import functional as Fn
dev = 'cuda:0'
c, rows = int(sys.argv[2]), int(sys.argv[3])
st = torch.rand(3, rows, c, device=dev) + 1
g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
for _ in range(300):
    Fn.bn_finalize(st, rows * 100, g, b, 1e-3, 0.1, rm, rv)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
for cfg in "64 512" "128 512" "256 128" "512 32" "512 150" "256 600" "64 4800"; do
  set -- $cfg
  rm -rf /tmp/bnp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/bnp -o p --output-format csv -- python /tmp/bnmicro.py $R $1 $2 > /dev/null 2>&1
  python - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/bnp/**/*kernel_stats.csv', recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    if 'bn_' in r['Name']:
        out.append(f"{r['Name'].split('(')[0]} {float(r['AverageNs'])/1e3:.1f} us")
print(f"c={sys.argv[1]} rows={sys.argv[2]}:", '; '.join(out))
PY
done
