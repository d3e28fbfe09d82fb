#!/bin/bash
for b in 768 512 1024 768 640; do echo "blocks $b"; EMSA_W1D_BLOCKS=$b bash tools/jobs/r04_wh3.sh base | tail -2; done
