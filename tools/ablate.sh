#!/bin/bash
# Lab build of the library with timing-only switches in the lock-step pass-1 kernel (-DPG_ABLATE: csrc/pair_gemm.hip PG_DBG), and a sweep of
# them over the dense-identity ceiling (tools/dense_ceiling.py) and the real rulebooks (tools/pair_quick.py).  The product library carries
# none of it.  usage: tools/ablate.sh build   (in the build container)   |   tools/ablate.sh run [out.md]   (on the GPU box)
set -e
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
if [ "$1" = "build" ]; then
    mkdir -p _ab
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DPG_ABLATE -DSD3D_WITH_PC -c segdino3d_amd/csrc/pair_gemm.hip -o _ab/pair_gemm_ablate.o
    objs=$(ls segdino3d_amd/csrc/*.o | grep -v pair_gemm.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _ab/pair_gemm_ablate.o -o _ab/libsegdino3d_hip_ablate.so
    exit 0
fi
out="${2:-gpurun_out/r06_ablate.md}"
: > "$out"
for dbg in ${ABLATE_SET:-0 1 2 4 8 3 7 15}; do
    echo "## SD3D_PG_ABLATE=$dbg (1 no epilogue stores, 2 gathers from row 0, 4 weights from chunk 0, 8 no step barrier, 16 stores into an L2-resident window, 32 pc: no staging writes of finished tiles)" >> "$out"
    SD3D_LIB="$ROOT/_ab/libsegdino3d_hip_ablate.so" SD3D_PG_ABLATE=$dbg CEIL_ONLY_LS=1 python tools/dense_ceiling.py >> "$out" 2>&1
    if [ -z "$ABLATE_DENSE_ONLY" ]; then
    SD3D_LIB="$ROOT/_ab/libsegdino3d_hip_ablate.so" SD3D_PG_ABLATE=$dbg PAIR_CHAINED=product PAIR_CASES=1,4,7,9,11 python tools/pair_quick.py >> "$out" 2>&1
    fi
done
