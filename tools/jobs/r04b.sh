#!/bin/bash
# 16-bit conv kernel: where does a launch's time go -- probe builds of conv_h.hip (EMSA_CONVH_DBG),
# built here on the GPU box, timed with tools/conv_bench16.py.  Results: DESIGN.md section 7.
O=gpurun_out/r04b; mkdir -p $O
L=emsanet_amd/lib
timeout 600 python tools/conv_bench16.py fwd > $O/base.txt 2>&1
for d in 1 2 3 4 5; do
  mkdir -p /tmp/dbg$d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -munsafe-fp-atomics -w -DEMSA_CONVH_DBG=$d \
      -c emsanet_amd/csrc/conv_h.hip -o /tmp/dbg$d/conv_h.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/conv_mfma.o $L/conv_wino.o /tmp/dbg$d/conv_h.o $L/pointwise.o \
      $L/loss.o $L/postproc.o $L/graph_tools.o -o /tmp/dbg$d/libemsanet_hip.so &&
  EMSA_LIB=/tmp/dbg$d/libemsanet_hip.so timeout 600 python tools/conv_bench16.py fwd > $O/dbg$d.txt 2>&1
done
python - <<'P'
rows={}
names=['base','dbg1','dbg2','dbg3','dbg4','dbg5']
for n in names:
    try: lines=open(f'gpurun_out/r04b/{n}.txt').read().splitlines()
    except OSError: continue
    for l in lines:
        f=l.split()
        if 'fwd' in f:
            i=f.index('fwd'); rows.setdefault(' '.join(f[:i]),{})[n]=f[i+1]
print('us per launch, bf16 forward, bs 32: 1 = output pass only, 2 = no stores, 3 = no MFMAs, 4 = no activation loads, 5 = no weight loads')
print('%-22s'%'shape',*['%8s'%n for n in names])
for k,v in rows.items(): print('%-22s'%k,*['%8s'%v.get(n,'-') for n in names])
P
