#!/bin/bash
# A/B: 64-channel vs 32-channel K steps in conv_h (variant built on the box)
O=gpurun_out/r02t; mkdir -p $O
make -s -C emsanet_amd/csrc OUT=/tmp/var32/libemsanet_hip.so OBJDIR=/tmp/var32 EXTRA="-DEMSA_CONVH_K=32" > $O/build32.log 2>&1; echo "build32 rc=$?"
EMSA_LIB=/tmp/var32/libemsanet_hip.so timeout 900 python -m pytest tests/test_ops16_gpu.py -m gpu -x -q -k "conv16 or stem16" > $O/tests_k32.log 2>&1; echo "k32 tests rc=$?"; tail -2 $O/tests_k32.log
for v in k64 k32 k64x k32x; do
  case $v in
    k64*) lib="" ;;
    k32*) lib="EMSA_LIB=/tmp/var32/libemsanet_hip.so" ;;
  esac
  env $lib timeout 600 python tools/conv_bench16.py fwd > $O/cb16_fwd_$v.txt 2>&1
  env $lib timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d = json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], [(k['kernel'][:14], k['avg_us']) for k in d['conv_kernels'][:1]])
PY
done
paste <(sed -n 4,15p $O/cb16_fwd_k64.txt | awk '{printf "%-22s %8s\n", $1" "$2" "$3, $5}') <(sed -n 4,15p $O/cb16_fwd_k32.txt | awk '{print $5}')
