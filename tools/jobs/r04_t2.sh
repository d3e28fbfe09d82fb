#!/bin/bash
O=gpurun_out/r04_t2; mkdir -p $O
for rs in 0 1; do
EMSA_CONV_RS=$rs timeout 3000 python -m pytest tests/test_model16_gpu.py -x -q -s -k "pinned and shape1" > $O/t_rs$rs.log 2>&1; echo "rs=$rs rc=$?"; grep "bf16 train:\|rel-L2 vs emul\|gradient norm ratio" $O/t_rs$rs.log | cut -c1-330
done
