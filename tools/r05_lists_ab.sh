#!/bin/bash
# chained builder in row-block form + workgroup-per-superpoint pooling: parity, then a one-stream kernel trace and two bench lines
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pair_paths.py tests/test_gpu_sparse.py tests/test_gpu_batch_eval.py tests/test_gpu_fullsize.py tests/test_gpu_real_sizes.py -m gpu -q > gpurun_out/lists_tests.txt 2>&1
tail -5 gpurun_out/lists_tests.txt
bash tools/profile_run.sh lists_trace_batch1 --steps 8 --warmup 2 --streams 1 --batch 1 --preroll-seconds 0.2 --no-end-to-end --sustain-seconds 0.2 > /dev/null 2>&1
grep -E "chain_|pair_(count|scan|fill|rowlist)|pool_superpoints|kmap|child_info|total kernel" gpurun_out/lists_trace_batch1.md | head -20
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/lists_bench.json 2> gpurun_out/lists_bench.err
python - <<'PY'
import json
for l in open("gpurun_out/lists_bench.json"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("value", d["value"], "single", (d.get("single_scene") or {}).get("latency_ms"), "frac", r["frac"], "conv ms", r["ms_per_forward"], "e2e", (d.get("end_to_end") or {}).get("value"))
PY
