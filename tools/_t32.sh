# parity tests of the FP32 strip kernels against one A/B build
VSM_LIB_PATH=$PWD/$1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "gauss_jordan or thick_conservative or fp32_strip or interaction or c4 or mixing" 2>&1 | tail -2
