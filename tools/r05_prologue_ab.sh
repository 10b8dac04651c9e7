#!/bin/bash
# prologue work: parity, same-process A/B of the switches at the single-scene operating point, host / GPU timeline, bench
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_batch_eval.py tests/test_gpu_fullsize.py tests/test_gpu_pair_paths.py tests/test_gpu_benchmark_parity.py -m gpu -q -x > gpurun_out/prologue_tests.txt 2>&1
tail -3 gpurun_out/prologue_tests.txt
AB_ONLY="fork / join" python tools/ab_single.py 8 2>&1 | grep median | tee gpurun_out/prologue_ab.txt
python tools/host_gpu_timeline.py 2>&1 | tail -9 | tee -a gpurun_out/prologue_ab.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prologue_bench.json 2> gpurun_out/prologue_bench.err
python - <<'PY'
import json
for l in open("gpurun_out/prologue_bench.json"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("value", d["value"], "single", (d.get("single_scene") or {}).get("latency_ms"), "frac", r["frac"], "conv ms", r["ms_per_forward"], "e2e", (d.get("end_to_end") or {}).get("value"))
PY
