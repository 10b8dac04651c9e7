#!/bin/bash
# The round's committed measurements in one go (GPU box): bench line, single-stream kernel trace, the two PMC passes, parity prints.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
bash tools/profile_run.sh final_trace --steps 8 --warmup 2 --streams 1 --preroll-seconds 0.2
PMC="FETCH_SIZE" bash tools/pmc_run.sh final_pmc_fetch --steps 4 --warmup 2 --streams 1 --preroll-seconds 0.2
PMC="WRITE_SIZE" bash tools/pmc_run.sh final_pmc_write --steps 4 --warmup 2 --streams 1 --preroll-seconds 0.2
python -m pytest tests/test_gpu_benchmark_parity.py tests/test_gpu_decoder.py tests/test_gpu_sparse.py tests/test_gpu_bf16_decoder.py -m gpu -s -q > gpurun_out/final_parity.txt 2>&1
tail -3 gpurun_out/final_parity.txt
