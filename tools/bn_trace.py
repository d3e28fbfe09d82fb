"""Durations of the BatchNorm-family kernels of one bf16 training step by launch size, from a
rocprofv3 --kernel-trace of `bench.py --dtype bf16` (one stream).  usage: python tools/bn_trace.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for pat in ('bn_act_fwd', 'bn_bwd_apply', 'bn_bwd_reduce', 'bn_bwd_sum', 'bn_finalize_rows', 'se_scale', 'channel_dot_kernel'):
    agg = collections.defaultdict(list)
    for r in rows:
        if pat in r['Kernel_Name']:
            agg[(int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']))].append(
                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print(pat)
    for g in sorted(agg):
        d = sorted(agg[g])
        print(f"   grid {g:9d}  launches {len(d):5d}  median {d[len(d) // 2]:7.1f} us  min {d[0]:7.1f}")
