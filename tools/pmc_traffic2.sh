#!/bin/bash
# GPU box: HBM traffic of the conv kernels from PMC counters (MI355X_MICROARCH.md, HBM section):
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (no trace domains beside --kernel-trace), each pass
# preceded by the three 1 GiB calibration kernels of tools/pmc_calib.hip in a pass of its own.
#   usage: tools/pmc_traffic2.sh <out-tag> [bench args ...]      -> gpurun_out/pmc_<tag>/raw.json
TAG=$1; shift
R=$PWD; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -o p -- $R/tools/bin/pmc_calib > $O/calib_$c.log 2>&1
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --roofline-steps 0 "$@" > $O/$c.log 2>&1
done
cd $R
python - "$O" <<'PY'
import collections, csv, glob, json, sys
O = sys.argv[1]
res = {}
for tag, sub in (('FETCH_SIZE', 'FETCH_SIZE'), ('WRITE_SIZE', 'WRITE_SIZE'),
                 ('calib_FETCH_SIZE', 'calib_FETCH_SIZE'), ('calib_WRITE_SIZE', 'calib_WRITE_SIZE')):
    c = tag.replace('calib_', '')
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f'{O}/{sub}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            agg[r['Kernel_Name']][0] += float(r['Counter_Value'])
            agg[r['Kernel_Name']][1] += 1
    res[tag] = {k: {'sum': v[0], 'launches': v[1]} for k, v in agg.items()}
json.dump(res, open(f'{O}/raw.json', 'w'))
for tag in ('calib_FETCH_SIZE', 'calib_WRITE_SIZE'):
    for k, v in res[tag].items():
        print(f"{tag:18s} {k[:60]:60s} launches {v['launches']:3d} avg {v['sum'] / v['launches']:14.1f} KiB")
PY
