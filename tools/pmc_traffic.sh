#!/bin/bash
# GPU box: HBM traffic of the conv kernels from PMC counters (MI355X_MICROARCH.md §HBM):
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes, calibrated on a known-size copy in the same pass.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_traffic
cat > /tmp/pmc_target.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
# calibration: 4 copies of exactly 1 GiB (read 1 GiB + write 1 GiB each), float4-wide
a = torch.empty(1 << 28, device='cuda:0'); b = torch.empty_like(a)
a.normal_()
for _ in range(4):
    b.copy_(a)
torch.cuda.synchronize()
sys.argv = ['bench.py', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-kernel-timing'] \
    + os.environ.get('EMSA_PMC_BENCH_ARGS', '').split()
exec(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'bench.py')).read())
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_traffic/$c -o p -- python /tmp/pmc_target.py > gpurun_out/pmc_traffic/$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f'gpurun_out/pmc_traffic/{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    res[c] = {k: (v[0], v[1]) for k, v in agg.items()}
out = {}
GIB = float(1 << 30)
# calibration kernel: the elementwise copy of 2^28 floats (largest "copy" entry, 4 launches + bench copies)
def find_copy(d):
    cands = [(k, v) for k, v in d.items() if 'copy' in k.lower() or 'direct_copy' in k]
    return cands
print('copy-like kernels:', {k: (v[0] / v[1], v[1]) for k, v in find_copy(res['FETCH_SIZE'])})
for c in res:
    for k, (tot, n) in sorted(res[c].items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"{c:11s} {k[:90]:90s} launches {n:6d} avg {tot / n:12.1f}")
json.dump({c: {k: {'sum': v[0], 'launches': v[1]} for k, v in d.items()} for c, d in res.items()},
          open('gpurun_out/pmc_traffic/raw.json', 'w'))
PY
