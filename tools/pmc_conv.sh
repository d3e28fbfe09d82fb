#!/bin/bash
# GPU box: PMC counters for the conv kernels on the micro-benchmark (separate passes per counter set)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_]+" | sort -u > gpurun_out/pmc/counters.txt
wc -l gpurun_out/pmc/counters.txt
what=${1:-fwd}
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_INT32 SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc/set$i -o p -- python tools/conv_bench.py $what -1 > gpurun_out/pmc/set$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc/set*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_' not in k:
            continue
        k = k.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        grid = r.get('Grid_Size', '')
        key = (k, grid)
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(key, r['Counter_Name'])] += 1
for key in sorted(agg):
    print(key)
    for c, v in sorted(agg[key].items()):
        print(f"   {c:32s} {v / cnt[(key, c)]:.4g}")
PY
