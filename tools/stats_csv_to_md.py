"""rocprofv3 `--kernel-trace --stats` CSV (p_kernel_stats.csv) -> markdown table.
usage: python tools/stats_csv_to_md.py p_kernel_stats.csv n_steps_in_the_run "title" > out.md"""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    title = sys.argv[3] if len(sys.argv) > 3 else sys.argv[1]
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f"# {title}\n")
    print(f"total GPU kernel time {tot / 1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} dispatches "
          f"({steps:g} steps incl. warm-up: {tot / 1e6 / steps:.2f} ms per step), {len(rows)} distinct kernels\n")
    print("| kernel | calls/step | ms/step | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for r in rows[:45]:
        n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
        n = re.sub(r'^void ', '', n)
        n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
        if len(n) > 100:
            n = n[:97] + '...'
        print(f"| `{n}` | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
              f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")


if __name__ == '__main__':
    main()
