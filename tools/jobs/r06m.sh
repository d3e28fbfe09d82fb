#!/bin/bash
# phase timers of conv_rs_kernel (register-staged loader, final sources of round 6): debug build in tools/bin/dbg
O=gpurun_out/r06m; mkdir -p $O
LD_LIBRARY_PATH=tools/bin/dbg timeout 600 tools/bin/conv_rs_probe 32 time > $O/phases.txt 2>&1; echo "rc=$?"; grep -v "^$" $O/phases.txt | cut -c1-260 | head -50
