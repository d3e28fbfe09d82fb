"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel stats table
(name, calls, total ms, avg us, min, max, % of GPU kernel time).
usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), "
                       f"max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |",
             "|---|---:|---:|---:|---:|---:|---:|"]
    for n, c, t, mn, mx in rows[:40]:
        n = re.sub(r'\(anonymous namespace\)::', '', n)
        n = re.sub(r'void ', '', n)
        if len(n) > 110:
            n = n[:107] + '...'
        lines.append(f"| `{n}` | {c} | {t / 1e6:.2f} | {t / c / 1e3:.1f} | {mn / 1e3:.1f} | "
                     f"{mx / 1e3:.1f} | {100 * t / total:.1f} |")
    lines.append(f"\ntotal GPU kernel time {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} "
                 f"dispatches, {len(rows)} distinct kernels")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out + "\n")


if __name__ == '__main__':
    main()
