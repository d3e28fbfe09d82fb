"""Per launch shape (grid size) time table of ONE kernel from a rocprofv3 rocpd database: which
layer shapes the dominant kernel spends its time on.
usage: python tools/rocpd_by_grid.py <results.db> <kernel name substring> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = f"%{sys.argv[2]}%"
    rows = db.execute(
        "select grid_x / workgroup_x, lds_size, count(*), sum(end-start), min(end-start), "
        "max(end-start) from kernels where name like ? group by 1, 2 order by 4 desc",
        (pat,)).fetchall()
    total = sum(r[3] for r in rows)
    lines = [f"kernel `{sys.argv[2]}`: {sum(r[2] for r in rows)} launches, {total / 1e6:.2f} ms",
             "", "| workgroups | LDS B | launches | total ms | avg us | min us | max us | % |",
             "|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for wg, lds, c, t, mn, mx in rows:
        lines.append(f"| {wg} | {lds} | {c} | {t / 1e6:.2f} | {t / c / 1e3:.1f} | {mn / 1e3:.1f} | "
                     f"{mx / 1e3:.1f} | {100 * t / total:.1f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write(out + "\n")


if __name__ == '__main__':
    main()
