#!/bin/bash
# phase timing of conv_rs (debug build with EMSA_RS_DBG=1 in tools/bin/dbg), then the product build's times
O=gpurun_out/r04_rs2; mkdir -p $O
LD_LIBRARY_PATH=tools/bin/dbg timeout 600 tools/bin/conv_rs_probe 32 time > $O/phases.txt 2>&1; echo "rc=$?"; cat $O/phases.txt
timeout 600 tools/bin/conv_rs_probe 32 time > $O/time.txt 2>&1; echo "time rc=$?"; cat $O/time.txt
