#!/bin/bash
# conv_h 256x128 eight-wave tile (EMSA_CONVH_TILE=3): op tests, then the 3x3 rows of conv_bench16, default vs forced
O=gpurun_out/r06p; mkdir -p $O
EMSA_CONVH_TILE=3 timeout 900 python -m pytest tests/test_ops16_gpu.py -m gpu -q -x -k "conv16 and not wgrad" > $O/ops16_tile3.log 2>&1; echo "ops16 tile3 rc=$?"; tail -2 $O/ops16_tile3.log
for t in default 1 3; do
  if [ $t = default ]; then unset EMSA_CONVH_TILE; else export EMSA_CONVH_TILE=$t; fi
  python tools/conv_bench16.py fwd 2>&1 | grep -E "3x3|1x1 c256|lib:" | awk -v t=$t '{print "tile", t, $0}' | cut -c1-120
  python tools/conv_bench16.py dgrad 2>&1 | grep -E "3x3" | awk -v t=$t '{print "tile", t, $0}' | cut -c1-120
done
