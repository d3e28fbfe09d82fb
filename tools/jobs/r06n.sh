#!/bin/bash
# conv_rs wave-tile variants (TN = 2: one A read per two MFMAs): correctness + time, default vs variant library
O=gpurun_out/r06n; mkdir -p $O
timeout 600 tools/bin/conv_rs_probe 32 time > $O/default_time.txt 2>&1; echo "default rc=$?"
LD_LIBRARY_PATH=tools/bin/var timeout 600 tools/bin/conv_rs_probe 32 all > $O/var_all.txt 2>&1; echo "var rc=$?"; tail -1 $O/var_all.txt
grep -E "^(1x3|3x1)" $O/default_time.txt | awk '{print "default", $0}' | cut -c1-110
grep -E "^(1x3|3x1)" $O/var_all.txt | awk '{print "variant", $0}' | cut -c1-110
grep -i "fail\|not sup" $O/var_all.txt | head
timeout 600 python -m pytest tests/test_parallel_gpu.py -m gpu -q -k "capture_and_fallback" 2>&1 | tail -3
