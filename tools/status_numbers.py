"""Print the numbers DESIGN.md section 0 quotes, straight from profiles/r06_z_*.json (one file per number;
profiles/INDEX.md is the map).  usage: python tools/status_numbers.py [prefix=profiles/r06_z_]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRE = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'r06_z_')


def load(name):
    try:
        return json.loads(open(PRE + name + '.json').read().strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        return {'_err': str(e)}


def main():
    for name in ('bench_driver_cmd_f32', 'bench_f32_eager', 'bench_bf16', 'bench_bf16_graph', 'bench_bf16_graph_2',
                 'bench_bf16_graph_bn1_fold16_off', 'bench_bf16_graph_bn1_fold16_everywhere',
                 'config3_r101_960x736_bs16_f32', 'config3_r101_960x736_bs16_bf16', 'config4_eval_graph_bs1_f16',
                 'config4_eval_graph_bs1_bf16', 'config4_eval_graph_bs1_f32', 'eval_bs32_f32', 'eval_bs32_bf16',
                 'bench_f32_forcedist_segmented_graph', 'bench_bf16_forcedist_segmented_graph'):
        d = load(name)
        if '_err' in d:
            print(f'{name}: {d["_err"]}')
            continue
        r = d.get('roofline') or {}
        ws = d.get('whole_step') or {}
        print(f"{name}: {d['value']} {d['unit']}  {d['ms_per_step']} ms  roofline {r.get('frac')} "
              f"({r.get('achieved')} {r.get('unit')}, traffic {r.get('traffic')})  hbm_frac {ws.get('hbm_frac')}  "
              f"nodes {(d.get('hipgraph') or {}).get('nodes')}")
        cb = d.get('cpu_baseline')
        if cb:
            print(f"   cpu_baseline {cb.get('value')} on {cb.get('cores')} threads, usable_cpus {cb.get('usable_cpus')}")
        c = d.get('comm') or {}
        if c.get('bucket_launch_ms_before_backward_end'):
            print(f"   bucket lead ms {c['bucket_launch_ms_before_backward_end']}  cu budget {c.get('conv_rs_cu_budget')}")
    rp = (load('config4_eval_graph_bs1_f16_reference_protocol').get('reference_protocol') or {})
    for k in ('as_reference', 'pinned_raw', 'pinned_compact'):
        v = rp.get(k) or {}
        print(f"reference protocol {k}: {v.get('fps_mean')} +- {v.get('fps_std')} FPS, {v.get('ms_mean')} ms, reps "
              f"{v.get('reps_fps_mean')}, D2H {v.get('device_to_host_bytes')}")


if __name__ == '__main__':
    main()
