#!/bin/bash
O=$PWD/gpurun_out/r04_wh3; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
python $R/tools/conv_bench16.py wgrad 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3}' | tr '\n' ';'; echo
for e in "$@"; do
  lib=$R/tools/bin/$e/libemsanet_hip.so; [ $e = base ] && lib=$R/emsanet_amd/lib/libemsanet_hip.so
  rm -rf $O/tr_$e
  EMSA_LIB=$lib timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$e -o p -- python $R/tools/conv_bench16.py wgrad > $O/tr_$e.log 2>&1
  echo "== $e"; python $R/tools/wgrad_trace.py $(find $O/tr_$e -name "*kernel_trace.csv" | head -1)
  rm -rf $O/tr_$e
done
