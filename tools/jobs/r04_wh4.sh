#!/bin/bash
O=$PWD/gpurun_out/r04_wh4; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
export EMSA_BENCH_SHAPE="${SHAPE:-1x3 c128}"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf $O/p$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/tools/conv_bench16.py wgrad > $O/p$i.log 2>&1
  python - $O/p$i <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wgrad1d_h' in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, v in agg.items():
    print(f"{k:28s} {v[0] / max(v[1], 1):16.0f}  ({v[1]} launches)")
PY
  rm -rf $O/p$i
done
