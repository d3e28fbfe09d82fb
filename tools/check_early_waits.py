"""Static check of the MFMA kernels' K loops: an `s_waitcnt vmcnt(k)` that sits between the global
loads of the next step and the MFMAs of the current one, with more than k loads issued since the
loop top, puts the wave to sleep on fresh data before its matrix work (seen twice: a bias sum
behind the load, a control-flow join behind the load).  Cross-compiles the sources and scans the
ISA; exit code 1 if such a wait exists.   usage: python tools/check_early_waits.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'emsanet_amd', 'csrc')


def isa(src):
    out = f'/tmp/_early_{os.path.basename(src)}.s'
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17',
                    f'-I{ROOT}/include', f'-I{CSRC}', '-munsafe-fp-atomics', '-S',
                    '--cuda-device-only', '-o', out, src], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    bad = 0
    for f in ('conv_mfma.hip', 'conv_wino.hip', 'conv_h.hip'):
        text = isa(os.path.join(CSRC, f))
        for m in re.finditer(r'^(_Z\S+):.*?s_endpgm', text, re.S | re.M):
            lines = m.group(0).split('\n')
            mf = [i for i, l in enumerate(lines) if re.match(r'\s+v_mfma', l)]
            if not mf:
                continue
            hdr = [i for i, l in enumerate(lines) if 'Loop Header: Depth=1' in l and i < mf[0]]
            if not hdr:
                continue
            name = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', m.group(1))[:60]
            # loop body = header .. last branch back to a label at or just before the header
            labels = [lines[j].split(':')[0] for j in range(max(0, hdr[-1] - 8), hdr[-1] + 1)
                      if re.match(r'\.LBB\d+_\d+:', lines[j])]
            back = [i for i, l in enumerate(lines) if i > hdr[-1] and
                    any(re.search(r's_c?branch\w*\s+' + re.escape(lb) + r'\s*$', l) for lb in labels)]
            if not back:
                continue
            body = lines[hdr[-1]:back[-1] + 1]
            loads = [i for i, l in enumerate(body) if re.search(r'\s(buffer|global)_load', l)]
            if not loads:
                continue
            cyc = body[loads[0]:] + body[:loads[0]]        # start at the first load, wrap around
            issued, mfma_since = 0, False
            for l in cyc:
                if re.search(r'\s(buffer|global)_load', l):
                    if mfma_since:
                        issued, mfma_since = 0, False
                    issued += 1
                elif re.match(r'\s+v_mfma', l):
                    mfma_since = True
                else:
                    w = re.search(r's_waitcnt.*vmcnt\((\d+)\)', l)
                    if w and issued > int(w.group(1)):
                        if not mfma_since:
                            print(f"{f}: {name}: '{l.strip()}' blocks on {issued} fresh loads "
                                  "before any MFMA of the step")
                            bad += 1
                        issued = int(w.group(1))
    print('early waits:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
