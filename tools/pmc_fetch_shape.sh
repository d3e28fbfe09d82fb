#!/bin/bash
# GPU box: HBM/MALL-side bytes (FETCH_SIZE x2 on gfx950, WRITE_SIZE) of the conv kernels for ONE
# layer shape of tools/conv_bench.py:  tools/pmc_fetch_shape.sh "<shape substring>" <what>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcf_$c
  EMSA_BENCH_SHAPE="$1" timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcf_$c -o p -- python tools/conv_bench.py ${2:-wino} > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f'/tmp/pmcf_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            if 'conv' in k or 'wino' in k:
                agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, (v, n) in agg.items():
        mb = v / n * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e6
        print(f"{c:11s} {k:45s} {n:4d} launches  {mb:8.1f} MB per launch")
PY
