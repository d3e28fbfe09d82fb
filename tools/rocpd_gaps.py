"""GPU idle time inside the profiled steps from a rocprofv3 rocpd database (kernel trace):
busy = union of kernel intervals, span = first start .. last end of the steady-state window.
usage: python tools/rocpd_gaps.py <results.db> [skip_fraction=0.4]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    rows = db.execute("select start, end from kernels order by start").fetchall()
    rows = rows[int(len(rows) * skip):]            # drop init + warm-up
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = t1 - t0
    gaps.sort(reverse=True)
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms "
          f"({100 * busy / span:.1f} %)  idle {(span - busy) / 1e6:.2f} ms")
    print(f"gaps: n={len(gaps)}  mean {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us  "
          f"top5 {[round(g / 1e3, 1) for g in gaps[:5]]} us  "
          f">10us: {sum(1 for g in gaps if g > 1e4)}  sum>10us {sum(g for g in gaps if g > 1e4) / 1e6:.2f} ms")


if __name__ == '__main__':
    main()
