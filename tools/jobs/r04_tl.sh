#!/bin/bash
# batch-1 inference graph: per-kernel timeline of one replay (two streams / one stream) + sanity of HEAD
O=gpurun_out/r04tl; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ds in 1 0; do
  EMSA_DUAL_STREAM=$ds timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_ds$ds -o p -- python $R/bench.py --eval --graph --batch-size 1 --dtype f16 --steps 60 --warmup 10 --no-cpu-baseline > $R/$O/bench_ds$ds.json 2> $R/$O/bench_ds$ds.err
  echo "trace ds=$ds rc=$?"; tail -1 $R/$O/bench_ds$ds.json | cut -c1-200
done
cd $R
for ds in 1 0; do
  f=$(ls $O/tr_ds$ds/*kernel_trace.csv 2>/dev/null | head -1)
  n=$(python -c "import json,sys; d=json.loads(open('$O/bench_ds$ds.json').read().strip().splitlines()[-1]); n=(d.get('hipgraph') or {}).get('nodes', 271); print(n[0] if isinstance(n, list) else n)")
  echo "ds=$ds nodes=$n file=$f"
  python tools/graph_timeline.py $f $n 2 > $O/timeline_ds$ds.txt 2>&1; tail -4 $O/timeline_ds$ds.txt
  rm -rf $O/tr_ds$ds
done
EMSA_DUAL_STREAM=1 timeout 300 python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 200 --warmup 20 --no-cpu-baseline > $O/b_ds1.json 2>/dev/null; cut -c1-160 $O/b_ds1.json | tail -1
EMSA_DUAL_STREAM=0 timeout 300 python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 200 --warmup 20 --no-cpu-baseline > $O/b_ds0.json 2>/dev/null; cut -c1-160 $O/b_ds0.json | tail -1
timeout 900 python -m pytest tests/test_ops16_gpu.py tests/test_conv_rs_gpu.py -m gpu -x -q > $O/tests16.log 2>&1; echo "tests16 rc=$?"; tail -2 $O/tests16.log
