#!/bin/bash
timeout 900 python -m pytest tests/test_ops16_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
echo "pd1 768"; bash tools/jobs/r04_wh3.sh base | tail -2
echo "pipe 768"; bash tools/jobs/r04_wh3.sh pipe | tail -2
echo "pd2 occ2 512"; EMSA_W1D_BLOCKS=512 bash tools/jobs/r04_wh3.sh pd2 | tail -2
echo "pd1 512"; EMSA_W1D_BLOCKS=512 bash tools/jobs/r04_wh3.sh base | tail -2
echo "pd1 768"; bash tools/jobs/r04_wh3.sh base | tail -2
