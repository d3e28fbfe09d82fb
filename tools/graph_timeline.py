"""Timeline of ONE replay of a whole-model hipGraph from a rocprofv3 kernel trace (csv): per kernel its
start offset, duration, queue and the idle time in front of it on its queue, then the totals (span,
sum of durations, union busy time, idle share per queue).  Answers "is the batch-1 graph a chain of
kernel durations or a chain of dispatch gaps".
usage: python tools/graph_timeline.py <kernel_trace.csv> <nodes per replay> [replay index from the end=1]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return name[:70]


def main():
    path, n = sys.argv[1], int(sys.argv[2])
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rows = rows[len(rows) - back * n: len(rows) - (back - 1) * n]
    t0 = int(rows[0]['Start_Timestamp'])
    last_end = {}
    tot_dur, busy, cur_e = 0, 0, None
    cur_s = None
    qidle = {}
    print(f"{'start us':>9} {'dur us':>7} {'queue':>6} {'gap(q) us':>9} {'gap(any) us':>11}  kernel")
    prev_any_end = None
    for r in rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        q = r.get('Queue_Id', '0')
        gq = (s - last_end[q]) / 1e3 if q in last_end else float('nan')
        ga = (s - prev_any_end) / 1e3 if prev_any_end is not None else float('nan')
        if q in last_end and s > last_end[q]:
            qidle[q] = qidle.get(q, 0) + s - last_end[q]
        last_end[q] = max(e, last_end.get(q, 0))
        prev_any_end = e if prev_any_end is None else max(prev_any_end, e)
        tot_dur += e - s
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {q:>6} {gq:9.1f} {ga:11.1f}  {short(r['Kernel_Name'])}")
    busy += cur_e - cur_s
    span = max(int(r['End_Timestamp']) for r in rows) - t0
    print(f"\nkernels {len(rows)}  span {span / 1e3:.1f} us  sum of durations {tot_dur / 1e3:.1f} us  "
          f"union busy {busy / 1e3:.1f} us ({100 * busy / span:.1f} % of the span)")
    for q, v in sorted(qidle.items()):
        cnt = sum(1 for r in rows if r.get('Queue_Id', '0') == q)
        print(f"queue {q}: {cnt} kernels, idle between its kernels {v / 1e3:.1f} us")


if __name__ == '__main__':
    main()
