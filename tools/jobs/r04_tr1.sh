#!/bin/bash
timeout 900 python -m pytest tests/test_ops16_gpu.py -x -q -k "wgrad" 2>&1 | tail -5
bash tools/jobs/r04_wh3.sh base | tail -2
EMSA_WGRAD16_TR=0 bash tools/jobs/r04_wh3.sh base | tail -2
bash tools/jobs/r04_wh3.sh base | tail -2
EMSA_WGRAD16_TR=0 bash tools/jobs/r04_wh3.sh base | tail -2
