"""ISA check of the 16-bit weight-gradient kernels in conv_mfma.hip (their LDS layouts are built
for specific instructions):
  conv_wgrad1d_h_kernel  -- no ds_read2_b64 / ds_write2_b64 (the load-store optimizer must stay off:
                            fused, the 8-byte fragment reads and transposing stores go back to the
                            32-bank rule and 8 cycles), v_permlane32_swap present in the 3-tap modes
  conv_wgrad1d_tr_kernel -- fragments by ds_read_b64_tr_b16 (32 per K step), no v_perm / v_alignbit
                            in the K loop, 16-byte LDS stores
usage: python tools/check_wgrad16_isa.py [-v]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels():
    src = os.path.join(ROOT, 'emsanet_amd', 'csrc', 'conv_mfma.hip')
    out = '/tmp/conv_mfma_wgrad16_check.s'
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'emsanet_amd', 'csrc'),
                    '-munsafe-fp-atomics', '-S', '--cuda-device-only', src, '-o', out],
                   check=True, capture_output=True)
    txt = open(out).read().split('\n')
    starts = [i for i, l in enumerate(txt) if l.startswith('_ZN') and l.rstrip().endswith(':')
              or (l.startswith('_ZN') and ': ' in l and '@' in l)]
    for k, i in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(txt)
        name = txt[i].split(':')[0]
        if 'conv_wgrad1d_h_kernel' in name or 'conv_wgrad1d_tr_kernel' in name:
            body = txt[i:end]
            stop = next((j for j, l in enumerate(body) if 's_endpgm' in l), len(body))
            yield name, body[:stop]


def count(body, op):
    return sum(1 for l in body if l.strip().startswith(op))


def main():
    bad, seen = [], 0
    for name, body in kernels():
        seen += 1
        c = {op: count(body, op) for op in ('ds_read2_b64', 'ds_write2_b64', 'ds_read_b64_tr_b16',
                                            'v_permlane32_swap', 'v_perm_b32', 'v_alignbit_b32',
                                            'ds_write_b128', 'v_mfma', 'scratch_')}
        if '-v' in sys.argv:
            print(name[-48:], c)
        if c['scratch_']:
            bad.append((name, 'scratch'))
        if 'conv_wgrad1d_h_kernel' in name:
            if c['ds_read2_b64'] or c['ds_write2_b64']:
                bad.append((name, 'fused 8-byte LDS accesses'))
            if c['v_mfma'] >= 12 and not c['v_permlane32_swap']:     # the 3-tap modes
                bad.append((name, 'no v_permlane32_swap'))
        else:
            if c['ds_read_b64_tr_b16'] < 32 or c['v_perm_b32'] or c['v_alignbit_b32'] or \
                    c['ds_write_b128'] < 4:
                bad.append((name, 'not the transposed-read form'))
    for b in bad:
        print('FAILED', b)
    print('wgrad16 isa check:', 'FAILED' if bad or seen < 4 else f'ok ({seen} kernels)')
    return 1 if bad or seen < 4 else 0


if __name__ == '__main__':
    sys.exit(main())
