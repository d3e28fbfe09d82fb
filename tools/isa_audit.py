"""Static audit of the gfx950 ISA of every kernel in emsanet_amd/csrc (build container, no GPU):
finds the two defects DESIGN.md 4.3 (round 5) describes in the streaming kernels --

  ser   vector-memory loads that are followed by `s_waitcnt vmcnt(0)` before the next load is issued
        (a load behind a branch / control-flow join: one memory round trip per load);
  rcp   v_rcp_* instructions (each integer division by a run-time value emits one or more;
        a 64-bit division ~8 and ~120 instructions in total);
  smulhi  s_mul_hi_u32 (scalar 64-bit multiplies / divisions: per-tile index decomposition on `long`).

  python tools/isa_audit.py [file.hip ...] [--min-ser N]     (default: all of csrc, N = 3)

The counts are per kernel (static, not per iteration); read the loop in the `.s` before acting on one.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'emsanet_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
LOAD = re.compile(r'\b(global_load|buffer_load|flat_load)')


_CXXFILT = next((c for c in ('/opt/rocm/lib/llvm/bin/llvm-cxxfilt', '/usr/bin/c++filt') if os.path.exists(c)), None)


def demangle(name):
    if _CXXFILT is None:
        return name
    r = subprocess.run([_CXXFILT, name], capture_output=True, text=True)
    return r.stdout.strip() or name


def audit(path, tmp):
    out = os.path.join(tmp, os.path.basename(path) + '.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-S',
                    '--cuda-device-only', f'-I{CSRC}', f'-I{ROOT}/include', path, '-o', out],
                   check=True, stderr=subprocess.DEVNULL)
    rows = []
    name, lines = None, []

    def close():
        if name is None:
            return
        ser = 0
        for i, line in enumerate(lines):
            if not LOAD.search(line):
                continue
            for nxt in lines[i + 1:i + 4]:
                if LOAD.search(nxt):
                    break
                if 's_waitcnt vmcnt(0)' in nxt:
                    ser += 1
                    break
        body = '\n'.join(lines)
        rows.append((ser, len(re.findall(r'v_rcp_(iflag_)?f32', body)), body.count('s_mul_hi_u32'),
                     os.path.basename(path), name))
    with open(out) as f:                       # one linear pass (the .s of pointwise.hip is 10 MB)
        for raw in f:
            m = re.match(r'^(_Z\w+):', raw)
            if m:
                name, lines = m.group(1), []
                continue
            if name is None:
                continue
            if '.end_amdhsa_kernel' in raw:
                close()
                name = None
                continue
            t = raw.strip()
            if t and not t.startswith(';'):
                lines.append(t)
    return rows


def main():
    argv = sys.argv[1:]
    min_ser = 3
    if '--min-ser' in argv:
        i = argv.index('--min-ser')
        min_ser = int(argv[i + 1])
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith('--')]
    files = args or sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            rows += audit(f, tmp)
    rows.sort(reverse=True)
    print("ser rcp smulhi file kernel")
    for ser, rcp, smul, f, name in rows:
        if ser >= min_ser or rcp >= 8 or smul >= 20:
            print(ser, rcp, smul, f, demangle(name)[:140])


if __name__ == '__main__':
    main()
