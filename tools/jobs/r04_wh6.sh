#!/bin/bash
for n in 8 16 32 64; do echo "batch $n"; EMSA_BENCH_N=$n bash tools/jobs/r04_wh3.sh head | tail -2; done
