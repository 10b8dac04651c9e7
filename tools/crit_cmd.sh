#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest ${1:-tests/test_gpu_criterion.py tests/test_gpu_train_ops.py} -m gpu -x -q -s 2>&1 | tail -30
