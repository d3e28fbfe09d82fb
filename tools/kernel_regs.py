"""Resource usage of the kernels in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage):
python tools/kernel_regs.py emsanet_amd/csrc/conv_rs.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Iinclude',
       '-Iemsanet_amd/csrc', '-munsafe-fp-atomics'] + sys.argv[3:] + ['-c', src, '-o', '/tmp/kernel_regs.o',
       '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    if 'error' in line:
        print(line)
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    name = re.sub(r'\(anonymous namespace\)::', '', name)[:110]
    print(f"{v.get('VGPRs', 0):4d} v {v.get('AGPRs', 0):4d} a {v.get('ScratchSize', 0):4d} scratch "
          f"{v.get('Occupancy', 0)} occ  {name}")
