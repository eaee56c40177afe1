#!/bin/bash
# AddressSanitizer on the library's HOST side: builds libesme_hip_asan.so (make ASAN=1: -fsanitize=address on the host
# half of every source -- argument validation, descriptor / workspace carving of esme_hip_forward, launch code; device code is untouched) and
# drives the C-ABI host tests through it with the ASan runtime preloaded: every entry point's validation path, bad arguments, struct layouts,
# exported symbols.  Runs WITHOUT a GPU.  (With one, the stock ROCm runtime does not start under the ASan runtime: its
# hsa_amd_memory_pool_allocate interceptor aborts with "out of memory" before the first launch -- profiles/r04_asan_host.txt.)
set -e
cd "$(dirname "$0")/.."
make -C esm-efficient_amd/csrc ASAN=1 -j8 > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
export ESME_HIP_LIB=$PWD/esm-efficient_amd/esme/libesme_hip_asan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1
nm -D $ESME_HIP_LIB | grep -c __asan_ | sed 's/^/__asan_ references in the library: /'
LD_PRELOAD=$RT python -m pytest tests/test_host_cpu.py -q -x 2>&1 | tail -3
