#!/bin/bash
# LDS bounds asserts (no compute-sanitizer on ROCm): builds libesme_hip_debug.so (make DEBUG=1: every LDS access of the
# GEMM and attention kernels checks its byte range against the workgroup's allocation and traps outside it) and drives the kernel-level
# and model-level GPU tests through it.  A trap aborts the launch, i.e. fails the test that issued it.  GPU box only.
set -e
cd "$(dirname "$0")/.."
make -C esm-efficient_amd/csrc DEBUG=1 -j8 > /dev/null
export ESME_HIP_LIB=$PWD/esm-efficient_amd/esme/libesme_hip_debug.so
python -m pytest tests/test_hip_kernels.py tests/test_exact_gpu.py tests/test_half_gpu.py tests/test_half_robust_gpu.py tests/test_fuzz_gpu.py tests/test_model_gpu.py tests/test_attn_sb_gpu.py tests/test_half_guard_gpu.py tests/test_attn_qp16_gpu.py -q -x -m gpu 2>&1 | tail -3
