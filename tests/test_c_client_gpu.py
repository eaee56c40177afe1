"""The C ABI without Python on the other side: examples/c_abi_demo.cpp (plain C++ + the HIP runtime, no torch, no ctypes) is compiled
against include/esme_hip.h, linked with libesme_hip.so and run: one attention block of the packed forward on a ragged batch, checked inside
the program against a float64 host computation of the same block ("extern C, plain pointers and sizes, no torch
types in the signatures")."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_client_links_the_c_abi_and_matches_its_own_float64_block(tmp_path):
    exe = str(tmp_path / 'c_abi_demo')
    libdir = os.path.join(ROOT, 'esm-efficient_amd', 'esme')
    build = subprocess.run(['/opt/rocm/bin/hipcc', '-O2', '-std=c++17', '--offload-arch=gfx950', '-I', os.path.join(ROOT, 'include'),
                            os.path.join(ROOT, 'examples', 'c_abi_demo.cpp'), '-L', libdir, '-lesme_hip', f'-Wl,-rpath,{libdir}', '-o', exe],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(run.stdout)
    assert run.returncode == 0, run.stdout + run.stderr[-3000:]
    assert 'c_abi_demo: OK' in run.stdout
