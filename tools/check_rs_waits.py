"""ISA check of conv_rs.hip: no s_waitcnt vmcnt(...) may sit between the first and the last MFMA of a
conv_rs_kernel instantiation (it would drain the next tile's LDS-DMA in every iteration); lists the
vmcnt waits of every instantiation's tile loop.  usage: python tools/check_rs_waits.py [-v]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels():
    src = os.path.join(ROOT, 'emsanet_amd', 'csrc', 'conv_rs.hip')
    out = '/tmp/conv_rs_check.s'
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17',
                    '-I' + os.path.join(ROOT, 'include'), '-munsafe-fp-atomics', '-S',
                    '--cuda-device-only', src, '-o', out], check=True, capture_output=True)
    txt = open(out).read().split('\n')
    starts = [i for i, l in enumerate(txt) if l.startswith('_ZN') and 'conv_rs_kernel' in l and ':' in l]
    for k, i in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(txt)
        yield txt[i].split(':')[0], txt[i:end]


def main():
    bad = 0
    for name, body in kernels():
        mf = [j for j, l in enumerate(body) if 'v_mfma' in l]
        inside = [l.strip() for j, l in enumerate(body)
                  if 's_waitcnt' in l and 'vmcnt' in l and mf[0] <= j <= mf[-1]]
        if inside:
            bad += 1
        if inside or '-v' in sys.argv:
            print(name[-60:], len(mf), 'MFMAs; vmcnt waits inside the MFMA sequence:', inside[:6])
    print('conv_rs wait check:', 'FAILED' if bad else 'ok')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
