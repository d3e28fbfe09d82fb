#!/bin/bash
# first run of the register-stationary conv: correctness vs conv_h, then time per shape
O=gpurun_out/r04_rs1; mkdir -p $O
timeout 600 tools/bin/conv_rs_probe 32 check > $O/check.txt 2>&1; echo "check rc=$?"; tail -5 $O/check.txt
grep -c " ok" $O/check.txt; grep "FAIL\|NOT SUP\|error" $O/check.txt | head -40
timeout 600 tools/bin/conv_rs_probe 32 time > $O/time.txt 2>&1; echo "time rc=$?"; cat $O/time.txt
