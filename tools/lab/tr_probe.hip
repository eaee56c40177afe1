// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds u16 element e at byte 2e (value e); lane l reads at byte address
// addr[l]; prints the four 16-bit values each lane receives.   hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const int* addr, unsigned short* out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned int a = (unsigned int)(size_t)lds + (unsigned int)addr[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v[0] & 0xffff; out[threadIdx.x * 4 + 1] = v[0] >> 16;
    out[threadIdx.x * 4 + 2] = v[1] & 0xffff; out[threadIdx.x * 4 + 3] = v[1] >> 16;
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int *d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof h_addr); hipMalloc(&d_out, sizeof h_out);
    for (int mode = 0; mode < 3; ++mode) {
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, r = (l & 15) >> 2, p = l & 3;
            // mode 0: canonical image, lane l reads 8 B at element 4*l; mode 1: rows of 128 B: group g -> rows 4g..4g+3,
            // lane (r, p) reads row 4g + r, elements 4p..4p+3; mode 2: lane-major, lane (l & 15) reads row l & 15? (elements 0..3)
            if (mode == 0) h_addr[l] = 8 * l;
            else if (mode == 1) h_addr[l] = (4 * g + r) * 128 + p * 8;
            else h_addr[l] = (l & 15) * 128 + g * 8;
        }
        hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr(elem) %4d -> %4d %4d %4d %4d", l, h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
            if (l % 2) printf("\n");
        }
    }
    return 0;
}
