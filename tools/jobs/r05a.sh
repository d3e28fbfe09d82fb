#!/bin/bash
# round 5, first GPU call: new tests of this round, activation-gradient comparison,
# the whole suite with per-test durations (suite budget), a baseline bench line
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_postprocessing.py "tests/test_model_gpu.py::test_graph_memset_repair_handles_chained_memsets" "tests/test_model_gpu.py::test_eval_after_stats_only_train_forward_sees_new_running_stats" "tests/test_model16_gpu.py::test_twin_launches_match_separate_launches" -m gpu -q -x > $O/new_tests.log 2>&1; echo "new tests rc=$?"; tail -5 $O/new_tests.log
timeout 900 python tools/actgrad_compare.py bf16 256 320 8 --out $O/actgrad_bf16_256x320_bs8.txt > $O/actgrad_bf16.log 2>&1; echo "actgrad bf16 rc=$?"; tail -3 $O/actgrad_bf16.log
timeout 900 python tools/actgrad_compare.py f32 256 320 8 --out $O/actgrad_f32_256x320_bs8.txt > $O/actgrad_f32.log 2>&1; echo "actgrad f32 rc=$?"; tail -3 $O/actgrad_f32.log
timeout 900 python tools/actgrad_compare.py bf16 256 320 8 --plain --out $O/actgrad_bf16_plain_256x320_bs8.txt > $O/actgrad_bf16_plain.log 2>&1; echo "actgrad bf16 plain rc=$?"
timeout 2400 python -m pytest tests -m gpu -q --durations=80 > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench rc=$?"; cut -c1-300 $O/bench_f32.json
timeout 600 python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 80 --warmup 20 --protocol reference --no-cpu-baseline > $O/eval_ref_protocol_f16.json 2> $O/eval_ref.err; echo "eval ref rc=$?"; python -c "
import json;d=json.loads(open('$O/eval_ref_protocol_f16.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline'],d['reference_protocol'])"
