"""gpurun_out/pmc_traffic/raw.json (tools/pmc_traffic.sh) -> profiles/<name>.json: HBM bytes per
launch of every conv kernel class.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
128-B requests as 64 B, so reads are doubled (MI355X_MICROARCH.md, HBM section) -- checked against
the 1 GiB calibration copies of the same pass.   usage: python tools/pmc_traffic_json.py raw.json out.json [commit]"""
import json
import os
import sys


def readable(name):
    """rocprofv3 leaves kernels whose template arguments contain __bf16 / _Float16 mangled (or
    demangles them into nonsense like 'bool _Accum'): name<ints..., type> from the Itanium string"""
    import re
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name)
    if m:
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        rest = name[m.end() + n:]
        args = []
        for tok in re.finditer(r'Li(\d+)E|Lb([01])E|DF16b|DF16_|f', rest.split('EEv')[0]):
            t = tok.group(0)
            args.append(tok.group(1) if t.startswith('Li') else
                        ('true' if tok.group(2) == '1' else 'false') if t.startswith('Lb') else
                        {'DF16b': 'bf16', 'DF16_': 'f16', 'f': 'float'}[t])
        return base + ('<' + ','.join(args) + '>' if args else '')
    if 'bool _Accum' in name:          # broken demangling of a __bf16 argument: keep the base name
        return name.split('<')[0] + '<bf16 instance:' + str(__import__('zlib').crc32(name.encode()) % 1000) + '>'
    return name


def merge(d):
    out = {}
    for k, v in d.items():
        r = readable(k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0])
        e = out.setdefault(r, {'sum': 0.0, 'launches': 0})
        e['sum'] += v['sum']
        e['launches'] += v['launches']
    return out


def csrc_sha16(root):
    """content hash of the kernel sources: bench.py refuses a traffic file measured on other kernels"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, 'emsanet_amd', 'csrc', '*.hip')) +
                    glob.glob(os.path.join(root, 'emsanet_amd', 'csrc', '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def main():
    raw = json.load(open(sys.argv[1]))
    fetch, write = merge(raw['FETCH_SIZE']), merge(raw['WRITE_SIZE'])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = sys.argv[3] if len(sys.argv) > 3 else 'unknown'
    GIB_KIB = float(1 << 20)
    calib_rows, f_fetch, f_write = [], 2.0, 1.0
    if 'calib_FETCH_SIZE' in raw:
        # three kernels of exactly 1 GiB each (tools/pmc_calib.hip): counter / true KiB
        cf, cw = merge(raw['calib_FETCH_SIZE']), merge(raw['calib_WRITE_SIZE'])
        for name, rd, wr in (('pmc_calib_read_1gib', 1, 0), ('pmc_calib_write_1gib', 0, 1),
                             ('pmc_calib_copy_1gib', 1, 1)):
            f = cf.get(name, {'sum': 0.0, 'launches': 1})
            w = cw.get(name, {'sum': 0.0, 'launches': 1})
            calib_rows.append({'kernel': name, 'true_read_KiB': int(rd * GIB_KIB),
                               'true_write_KiB': int(wr * GIB_KIB),
                               'FETCH_SIZE_KiB_per_launch': round(f['sum'] / max(f['launches'], 1), 1),
                               'WRITE_SIZE_KiB_per_launch': round(w['sum'] / max(w['launches'], 1), 1),
                               'launches': f['launches']})
        rd_rows = [r for r in calib_rows if r['true_read_KiB'] and r['FETCH_SIZE_KiB_per_launch'] > 0]
        wr_rows = [r for r in calib_rows if r['true_write_KiB'] and r['WRITE_SIZE_KiB_per_launch'] > 0]
        if rd_rows:
            f_fetch = sum(r['true_read_KiB'] / r['FETCH_SIZE_KiB_per_launch'] for r in rd_rows) / len(rd_rows)
        if wr_rows:
            f_write = sum(r['true_write_KiB'] / r['WRITE_SIZE_KiB_per_launch'] for r in wr_rows) / len(wr_rows)
    out = {'commit': commit, 'csrc_sha16': csrc_sha16(root),
           'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- '
                      'python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing '
                      + os.environ.get('EMSA_PMC_BENCH_ARGS', '') +
                      ' ; calibration: the same two passes over tools/bin/pmc_calib',
           'calibration': {
               'rows': calib_rows,
               'fetch_factor': round(f_fetch, 4), 'write_factor': round(f_write, 4),
               'note': 'true bytes / counter over the 1 GiB read / write / copy kernels of '
                       'tools/pmc_calib.hip (16 bytes per lane, buffers rotating over 4 GiB); the '
                       'factors multiply FETCH_SIZE / WRITE_SIZE of the conv kernels below '
                       '(MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B on gfx950)'},
           'kernels': {}}
    for k in sorted(fetch):
        if not (k.startswith('conv') or 'wino' in k):
            continue
        n = fetch[k]['launches']
        rd = fetch[k]['sum'] / n * 1024 * f_fetch
        wr = write.get(k, {'sum': 0, 'launches': 1})
        wr = wr['sum'] / max(wr['launches'], 1) * 1024 * f_write
        out['kernels'][k.replace(' ', '')] = {
            'launches_profiled': n, 'hbm_read_bytes_per_launch': int(rd),
            'hbm_write_bytes_per_launch': int(wr), 'hbm_bytes_per_launch': int(rd + wr)}
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
    print('calibration:', json.dumps(out['calibration']['rows']), 'factors', f_fetch, f_write)
    for k, v in out['kernels'].items():
        print(f"{k:50s} {v['launches_profiled']:6d} launches  rd {v['hbm_read_bytes_per_launch'] / 1e6:8.1f} MB"
              f"  wr {v['hbm_write_bytes_per_launch'] / 1e6:8.1f} MB")


if __name__ == '__main__':
    main()
