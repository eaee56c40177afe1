#!/usr/bin/env python
"""Write a copy of csrc/attn.hip with block-level cycle stamps (entry, loop start, loop end, exit of wave 1 of every 8th
workgroup of attn_pp64_kernel) for tools/attn_trace.py:  make_attn_trace.py OUT.hip ; tools/lab/build_alt.sh attn.hip OUT.hip"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, 'esm-efficient_amd/csrc/attn.hip')).read()
for h in ('launch.h', 'gemm.h', 'common.h'):
    s = s.replace('#include "%s"' % h, '#include "../%s"' % h)


def sub(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


sub('template <int NW>\n__global__ __launch_bounds__(NW * 64, 2) void attn_pp64_kernel',
    '__device__ unsigned long long g_attn_trace[4096];\n'
    '#define TSTAMP(i) do { if (trace_on && lane == 0) { g_attn_trace[tix * 8 + (i)] = __builtin_readcyclecounter(); '
    'g_attn_trace[tix * 8 + 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } } while (0)\n\n'
    'template <int NW>\n__global__ __launch_bounds__(NW * 64, 2) void attn_pp64_kernel')
sub('    const unsigned int xcd = blockIdx.x & 7u, bi = blockIdx.x >> 3;\n    const int qt = (int)(bi % (unsigned int)a.nqt);',
    '    const bool trace_on = (blockIdx.x % 8u) == 0 && wave == 1 && blockIdx.x / 8u < 512;\n    const int tix = blockIdx.x / 8u;\n    TSTAMP(0);\n'
    '    const unsigned int xcd = blockIdx.x & 7u, bi = blockIdx.x >> 3;\n    const int qt = (int)(bi % (unsigned int)a.nqt);')
sub('        for (int t = 0; t < nt; ++t) {\n', '        TSTAMP(1);\n        for (int t = 0; t < nt; ++t) {\n')
sub('    if (!wave_active) return;\n\n    // ---- epilogue', '    TSTAMP(2);\n    if (!wave_active) return;\n\n    // ---- epilogue')
sub('    }\n}\n\n}  // namespace esme', '    }\n    TSTAMP(3);\n}\n\n}  // namespace esme')
s = s.rstrip() + '''

extern "C" int esme_hip_debug_read_attn_trace(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(esme::g_attn_trace), sizeof(unsigned long long) * 4096);
}
'''
open(sys.argv[1], 'w').write(s)
