#!/bin/bash
# round 3, first call: blockDim-under-graph-replay probe (stand-alone + in the library), ADVICE fixes,
# full GPU suite, baseline bench of this box for default vs code-object-v4 builds
O=gpurun_out/r03a; mkdir -p $O
R=$GRAFT_REPO_ROOT
echo "== stand-alone probe, code object v5"; timeout 120 tools/bin/repro_cov5 4000 2>&1 | tail -8
echo "== stand-alone probe, code object v4"; timeout 120 tools/bin/repro_cov4 4000 2>&1 | tail -8
T="tests/test_model_gpu.py::test_hipgraph_train_step_matches_eager"
for v in bd5 bd4 cov4; do
  echo "== hipGraph train step, library variant $v"
  EMSA_LIB=$R/emsanet_amd/lib/$v/libemsanet_hip.so timeout 600 python -m pytest $T -x -q 2>&1 | tail -4
done
echo "== full GPU suite, default library"
timeout 2700 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_gpu.log
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'))
except Exception as e:
    print('$name failed', e)
PY
}
run bench_f32 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
EMSA_LIB=$R/emsanet_amd/lib/cov4/libemsanet_hip.so run bench_f32_cov4 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
EMSA_LIB=$R/emsanet_amd/lib/cov4/libemsanet_hip.so run bench_bf16_cov4 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
run bench_f32_b --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
