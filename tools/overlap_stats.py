"""Concurrency inside ONE replay of a training-step hipGraph from a rocprofv3 kernel trace (csv): span,
sum of durations, union busy time, time with >= 2 kernels resident, and per kernel class the average
duration when the kernel ran alone vs overlapped with another kernel.
usage: python tools/overlap_stats.py <kernel_trace.csv> <kernels per replay> [replay index from the end=1]"""
import csv
import re
import sys
from collections import defaultdict


def cls(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:_ZN\d+_GLOBAL__N_1\d+)?([A-Za-z_0-9]+?)(?:_kernel)?(?:I|<|\(|$)', name)
    return (m.group(1) if m else name)[:40]


def main():
    path, n = sys.argv[1], int(sys.argv[2])
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rows = rows[len(rows) - back * n: len(rows) - (back - 1) * n]
    ev = []
    for i, r in enumerate(rows):
        ev.append((int(r['Start_Timestamp']), 1, i))
        ev.append((int(r['End_Timestamp']), -1, i))
    ev.sort()
    t0, t1 = ev[0][0], ev[-1][0]
    active, last = set(), t0
    busy = multi = 0
    alone_t = defaultdict(int)          # per kernel: time it was the only resident kernel
    for t, d, i in ev:
        dt = t - last
        if active:
            busy += dt
            if len(active) >= 2:
                multi += dt
            else:
                alone_t[next(iter(active))] += dt
        last = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
    queues = sorted(set(r.get('Queue_Id', '0') for r in rows))
    print(f"kernels {len(rows)}  queues {len(queues)}  span {(t1 - t0) / 1e3:.1f} us  sum of durations "
          f"{tot / 1e3:.1f} us  union busy {busy / 1e3:.1f} us  >= 2 kernels resident {multi / 1e3:.1f} us "
          f"({100.0 * multi / max(1, busy):.1f} % of busy)")
    agg = defaultdict(lambda: [0, 0, 0, 0, 0])      # count, dur, alone-count, alone-dur, overlapped-dur
    for i, r in enumerate(rows):
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg[cls(r['Kernel_Name'])]
        a[0] += 1
        a[1] += d
        if alone_t[i] >= 0.9 * d:
            a[2] += 1
            a[3] += d
        else:
            a[4] += d
    print(f"{'class':40s} {'calls':>6} {'ms':>8} {'avg us':>8} {'alone: n':>9} {'avg us':>8} {'overlapped: n':>13} {'avg us':>8}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        no = a[0] - a[2]
        print(f"{k:40s} {a[0]:6d} {a[1] / 1e6:8.3f} {a[1] / a[0] / 1e3:8.1f} {a[2]:9d} "
              f"{(a[3] / a[2] / 1e3 if a[2] else 0):8.1f} {no:13d} {(a[4] / no / 1e3 if no else 0):8.1f}")


if __name__ == '__main__':
    main()
