python -m pytest tests/test_gpu_slab_conv.py -m gpu -q -x 2>&1 | tail -3
python tools/slab_quick.py 2>&1 | grep -v amdgpu.ids
