#!/bin/bash
# The round's committed measurements in one go (GPU box): bench lines of every configuration, single-stream kernel trace, the two PMC
# traffic passes (keyed to the workload), parity prints.   usage: [ONLY_BENCH=1] tools/final_measure.sh [prefix, default r06]   (ONLY_BENCH: the bench lines, no traces / counter passes / parity prints)
P=${1:-r06}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out
python bench.py > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err
python bench.py --no-cpu-baseline --no-end-to-end --decoder-dtype bf16 > gpurun_out/${P}_bench_bf16.json 2>> gpurun_out/${P}_bench.err
python bench.py --no-cpu-baseline --no-end-to-end --query-num -1 > gpurun_out/${P}_bench_qall.json 2>> gpurun_out/${P}_bench.err
python bench.py --no-cpu-baseline --no-end-to-end --query-num -1 --decoder-dtype bf16 > gpurun_out/${P}_bench_qall_bf16.json 2>> gpurun_out/${P}_bench.err
python bench.py --no-cpu-baseline --no-end-to-end --batch 1 > gpurun_out/${P}_bench_batch1.json 2>> gpurun_out/${P}_bench.err
python bench.py --no-cpu-baseline --no-end-to-end --scene-layout scan > gpurun_out/${P}_bench_scan_layout.json 2>> gpurun_out/${P}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${P}_bench_driverline.json 2>> gpurun_out/${P}_bench.err
python bench.py --no-cpu-baseline --no-end-to-end --points 500000 --superpoints 10000 --steps 16 > gpurun_out/${P}_bench_500k.json 2>> gpurun_out/${P}_bench.err
python bench.py --no-cpu-baseline --no-end-to-end --points 500000 --superpoints 10000 --steps 16 --query-num -1 > gpurun_out/${P}_bench_500k_qall.json 2>> gpurun_out/${P}_bench.err
SD3D_DIST_BACKEND=gloo SD3D_SHARE_GPU=1 python bench.py --gpus 8 --steps 16 --warmup 2 --streams 2 --batch 2 --no-cpu-baseline --no-end-to-end --preroll-seconds 0.5 > gpurun_out/${P}_bench_8ranks_shared_gpu.json 2>> gpurun_out/${P}_bench.err
if [ -z "$ONLY_BENCH" ]; then
bash tools/profile_run.sh ${P}_kernel_trace_1stream_batch1 --steps 8 --warmup 2 --streams 1 --batch 1 --preroll-seconds 0.2 --no-end-to-end --sustain-seconds 0.2
bash tools/profile_run.sh ${P}_kernel_trace_1stream_batch4 --steps 8 --warmup 2 --streams 1 --batch 4 --preroll-seconds 0.2 --no-end-to-end --sustain-seconds 0.2
# the driver's forward sizes (20 steps on 4 streams x batches of <= 5: one forward of 5 scenes per stream), one stream, NO counters: the trace the roofline
# can be recomputed from (pair_gemm_* + pair_reduce_rl_kernel = the 55 counted convolutions; the U-Net's 1x1 convolutions are pair_dense_kernel_*)
bash tools/profile_run.sh ${P}_kernel_trace_1stream_fwd5 --steps 20 --warmup 5 --streams 1 --forward-sizes 5 --skip-single-scene --preroll-seconds 0.2 --no-end-to-end --sustain-seconds 0.1
bash tools/profile_run.sh ${P}_kernel_trace_1stream_qall --steps 8 --warmup 2 --streams 1 --batch 1 --preroll-seconds 0.2 --no-end-to-end --sustain-seconds 0.2 --query-num -1
# counter passes on exactly the forward sizes the driver's command line times (20 steps on 4 streams x batches of <= 5: forwards of 5 scenes),
# replayed on one stream; no one-scene forward anywhere in the profiled process
PMC="FETCH_SIZE" bash tools/pmc_run.sh ${P}_pmc_fetch --steps 20 --warmup 5 --streams 1 --forward-sizes 5 --skip-single-scene --preroll-seconds 0.2 --no-end-to-end --sustain-seconds 0.1
PMC="WRITE_SIZE" bash tools/pmc_run.sh ${P}_pmc_write --steps 20 --warmup 5 --streams 1 --forward-sizes 5 --skip-single-scene --preroll-seconds 0.2 --no-end-to-end --sustain-seconds 0.1
python tools/pmc_traffic.py gpurun_out/${P}_pmc_fetch.md gpurun_out/${P}_pmc_write.md gpurun_out/${P}_pmc_traffic.json gpurun_out/${P}_pmc_fetch.bench.json > /dev/null
python -m pytest tests/test_gpu_benchmark_parity.py tests/test_gpu_real_sizes.py tests/test_gpu_rowchain.py tests/test_gpu_batch_eval.py tests/test_gpu_pair_paths.py tests/test_gpu_decoder.py tests/test_gpu_sparse.py tests/test_gpu_bf16_decoder.py -m gpu -s -q > gpurun_out/${P}_parity.txt 2>&1
tail -3 gpurun_out/${P}_parity.txt
fi
for f in gpurun_out/${P}_bench*.json; do python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print(sys.argv[1].split("/")[-1], "value", d["value"], "n_gpus", d["n_gpus"], "single", (d.get("single_scene") or {}).get("latency_ms"), "frac", r["frac"], "conv ms", r["ms_per_forward"], "traffic x", r.get("traffic_over_algorithmic"), "e2e", (d.get("end_to_end") or {}).get("value"))
PY
done
