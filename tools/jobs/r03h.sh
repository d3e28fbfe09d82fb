#!/bin/bash
# evidence for DESIGN.md section 7 (round 3 probes): conv_h time decomposition, L2 / beyond-L2 read
# bandwidth, split-bf16 K loop
O=gpurun_out/r03h; mkdir -p $O
bash tools/jobs/r04b.sh > $O/conv_h_time_probes.txt 2>&1; tail -8 $O/conv_h_time_probes.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/l2_peak tools/l2_read_peak.hip && /tmp/l2_peak > $O/l2_read_peak.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/split_peak tools/split_bf16_peak.hip && /tmp/split_peak > $O/split_bf16_peak.txt 2>&1
tail -3 $O/l2_read_peak.txt; tail -3 $O/split_bf16_peak.txt
