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


def main():
    raw = json.load(open(sys.argv[1]))
    fetch, write = merge(raw['FETCH_SIZE']), merge(raw['WRITE_SIZE'])
    # calibration: the elementwise copies of exactly 1 GiB
    calib = {}
    for name, d in (('FETCH', fetch), ('WRITE', write)):
        best = None
        for k, v in d.items():
            if 'copy' not in k.lower():
                continue
            avg = v['sum'] / v['launches']
            if best is None or avg > best[1]:
                best = (k, avg, v['launches'])
        calib[name] = best
    commit = sys.argv[3] if len(sys.argv) > 3 else 'unknown'
    out = {'commit': commit, 'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- '
                      'python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing '
                      + os.environ.get('EMSA_PMC_BENCH_ARGS', '') + ' (after 4 calibration copies of 1 GiB)',
           'calibration': {
               'copy_bytes': 1 << 30,
               'copy_kernel': calib['FETCH'][0][:80],
               'FETCH_SIZE_KiB_avg_over_copy_launches': round(calib['FETCH'][1], 1),
               'WRITE_SIZE_KiB_avg_over_copy_launches': round(calib['WRITE'][1], 1),
               'note': 'gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled '
                       '(MI355X_MICROARCH.md HBM section); WRITE_SIZE exact; the copy class also '
                       'contains smaller copies of the bench, so its average is below 1 GiB'},
           'kernels': {}}
    for k in sorted(fetch):
        if not (k.startswith('conv') or 'wino' in k):
            continue
        n = fetch[k]['launches']
        rd = fetch[k]['sum'] / n * 1024 * 2
        wr = write.get(k, {'sum': 0, 'launches': 1})
        wr = wr['sum'] / max(wr['launches'], 1) * 1024
        out['kernels'][k.replace(' ', '')] = {
            'launches_profiled': n, 'hbm_read_bytes_per_launch': int(rd),
            'hbm_write_bytes_per_launch': int(wr), 'hbm_bytes_per_launch': int(rd + wr)}
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
    for k, v in out['kernels'].items():
        print(f"{k:50s} {v['launches_profiled']:6d} launches  rd {v['hbm_read_bytes_per_launch'] / 1e6:8.1f} MB"
              f"  wr {v['hbm_write_bytes_per_launch'] / 1e6:8.1f} MB")


if __name__ == '__main__':
    main()
