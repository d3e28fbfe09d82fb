#!/bin/bash
O=gpurun_out/r04_rs3; mkdir -p $O
timeout 600 tools/bin/conv_rs_probe 32 check > $O/check.txt 2>&1; echo "check rc=$?"; tail -1 $O/check.txt; grep "FAIL\|NOT SUP\|error" $O/check.txt | head -20
LD_LIBRARY_PATH=tools/bin/dbg timeout 600 tools/bin/conv_rs_probe 32 time > $O/phases.txt 2>&1; echo "rc=$?"; grep -A1 phases $O/phases.txt | grep -v "^--" | cut -c1-230
timeout 600 tools/bin/conv_rs_probe 32 time > $O/time.txt 2>&1; echo "time rc=$?"; cat $O/time.txt
