"""Summarise gpurun_out/pmc (tools/pmc_conv.sh passes over tools/conv_bench.py) into a markdown
table: matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)), LDS
bank-conflict share, VALU / LDS instructions per MFMA and wave wait share, per kernel.
usage: python tools/pmc_summary.py [gpurun_out/pmc] > profiles/<name>.md"""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(f'{root}/set*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            k = k.split('(')[0]
            if not (k.startswith('conv') or 'wino' in k) or 'pack' in k:
                continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(k, r['Counter_Name'])] += 1
    print('| kernel | MFMA pipe busy | LDS bank-conflict cycles | VALU inst / MFMA | '
          'LDS inst / MFMA | wave cycles waiting on an instruction |')
    print('|---|---:|---:|---:|---:|---:|')
    for k in sorted(agg):
        a = {c: v / cnt[(k, c)] for c, v in agg[k].items()}
        mf = a.get('SQ_INSTS_VALU_MFMA_F32', 0)
        if not mf:
            continue
        util = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 4 / max(a.get('SQ_BUSY_CU_CYCLES', 1), 1)
        conf = a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1)
        print(f"| `{k}` | {100 * util:.1f} % | {100 * conf:.0f} % of LDS-active | "
              f"{a.get('SQ_INSTS_VALU', 0) / mf:.1f} | {a.get('SQ_INSTS_LDS', 0) / mf:.2f} | "
              f"{100 * a.get('SQ_WAIT_INST_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.0f} % |")


if __name__ == '__main__':
    main()
