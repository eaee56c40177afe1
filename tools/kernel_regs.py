#!/usr/bin/env python
"""Register / LDS / spill table of every gfx950 kernel in a built object (csrc/build/*.o) or library:
    python tools/kernel_regs.py esm-efficient_amd/csrc/build/gemm.o [filter]
Unbundles the gfx950 code object from the fat binary and reads the AMDGPU metadata notes."""
import os, re, subprocess, sys, tempfile
LLVM = '/opt/rocm/lib/llvm/bin'

def kernels(path):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, 'fat.bin'), os.path.join(td, 'k.co')
        subprocess.run([f'{LLVM}/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', path, fat], check=True)
        subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}',
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'], check=True)
        notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], capture_output=True, text=True, check=True).stdout
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r'\s+-?\s*\.(\w+):\s+(.*)$', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == 'agpr_count' and cur.get('name'):
            out.append(cur); cur = {}
        if k in ('name', 'vgpr_count', 'agpr_count', 'vgpr_spill_count', 'sgpr_count', 'sgpr_spill_count', 'group_segment_fixed_size', 'private_segment_fixed_size'):
            cur[k] = v
    if cur.get('name'):
        out.append(cur)
    return out

if __name__ == '__main__':
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    for k in kernels(sys.argv[1]):
        name = subprocess.run(['c++filt', k['name']], capture_output=True, text=True).stdout.strip()
        if flt and flt not in name:
            continue
        print(f"vgpr {k.get('vgpr_count','?'):>4} agpr {k.get('agpr_count','?'):>4} spill {k.get('vgpr_spill_count','?'):>4} scratch {k.get('private_segment_fixed_size','?'):>5} lds {k.get('group_segment_fixed_size','?'):>6}  {name[:230]}")
