"""Rebuilds the HIP sources with -Rpass-analysis=kernel-resource-usage into a scratch directory,
prints registers / occupancy of every kernel and exits 1 if any kernel spills (a spilled prefetch
register puts an `s_waitcnt vmcnt(0)` right behind the global load and serialises the pipeline:
measured 158 us -> 230 us on the 1-D weight-gradient kernel).   usage: python tools/check_spills.py"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERNS = {'vgpr': r' VGPRs: (\d+)', 'agpr': r'AGPRs: (\d+)', 'spill': r'VGPRs Spill: (\d+)',
            'scratch': r'ScratchSize \[bytes/lane\]: (\d+)',
            'occ': r'Occupancy \[waves/SIMD\]: (\d+)'}


def main():
    out = tempfile.mkdtemp()
    try:
        extra = '-Rpass-analysis=kernel-resource-usage ' + os.environ.get('EXTRA', '')
        r = subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'emsanet_amd', 'csrc'),
                            f'OUT={out}/lib.so', f'OBJDIR={out}', f'EXTRA={extra}'],
                           capture_output=True, text=True)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    if r.returncode:
        print(r.stderr[-2000:])
        return 2
    rows, cur = [], None
    for line in (r.stdout + r.stderr).splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        for key, pat in PATTERNS.items():
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    bad = 0
    for row in rows:
        name = subprocess.run(['c++filt', row['name']], capture_output=True,
                              text=True).stdout.strip()
        name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        spills = row.get('spill', 0) or row.get('scratch', 0)
        bad += bool(spills)
        print(f"{name[-64:]:64s} vgpr {row.get('vgpr', 0):3d} agpr {row.get('agpr', 0):3d} "
              f"waves/SIMD {row.get('occ', 0)} scratch {row.get('scratch', 0)}"
              + ('   <-- SPILLS' if spills else ''))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
