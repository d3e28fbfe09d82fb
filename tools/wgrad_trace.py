"""Per-shape GPU durations of the 16-bit weight-gradient kernels from a rocprofv3 --kernel-trace of
`python tools/conv_bench16.py wgrad` (whose event timing includes the host's launch path: ~40 us per
call, more than most of these kernels take).   usage: python tools/wgrad_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = 27                                             # conv_bench.timeit: 3 warm-up + 24 timed calls
for pat in ("conv_wgrad1d_", "wgrad1d_reduce"):
    sel = sorted((r for r in rows if pat in r['Kernel_Name']), key=lambda r: int(r['Start_Timestamp']))
    out = []
    for i in range(0, len(sel) - per + 1, per):
        d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in sel[i + 3:i + per]]
        out.append(f"{sum(d) / len(d):6.1f}")
    print(f"{pat:24s}", ' '.join(out))
