// A client of libesme_hip.so that is NOT Python and knows nothing about torch: plain C++ + the HIP runtime for device memory, the C ABI of
// include/esme_hip.h for the arithmetic.  One attention block of the packed forward on a ragged batch --
//     h = LayerNorm(x);  qkv = h Wqkv^T + b (fused q/k/v);  rotary on q, k;  o = varlen attention;  y = x + o Wo^T + bo
// -- checked against a naive float64 CPU computation of the same block on the same bf16 inputs (the reference's
// FlashMultiheadAttention.forward, esme/attention.py:91-139, on the reference's unpadded layout).  Build and run (GPU box):
//     hipcc -O2 -std=c++17 -I include examples/c_abi_demo.cpp -L esm-efficient_amd/esme -lesme_hip -Wl,-rpath,$PWD/esm-efficient_amd/esme -o /tmp/c_abi_demo && /tmp/c_abi_demo
// tests/test_c_client_gpu.py does exactly that.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "esme_hip.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define ESME(x) do { int rc_ = (x); if (rc_ != ESME_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, esme_hip_last_error()); return 3; } } while (0)

static uint16_t f2bf(float f) {                       // round to nearest even
    uint32_t u; std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

struct Rng { uint64_t s; float next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 40) / 16777216.0f) * 2.0f - 1.0f; } };

template <class T> static T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main() {
    if (esme_hip_abi_version() != ESME_HIP_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    const int H = 4, d = 64, E = H * d;
    const std::vector<int> lengths = {37, 5, 130, 64};
    std::vector<int32_t> cu = {0};
    for (int L : lengths) cu.push_back(cu.back() + L);
    const int B = (int)lengths.size(), T = cu.back();
    int max_len = 0;
    for (int L : lengths) max_len = L > max_len ? L : max_len;

    Rng rng{12345};
    auto fill = [&](size_t n, float scale, float shift = 0.f) { std::vector<uint16_t> v(n); for (auto& e : v) e = f2bf(rng.next() * scale + shift); return v; };
    std::vector<uint16_t> x = fill((size_t)T * E, 1.5f), gamma = fill(E, 0.2f, 1.0f), beta = fill(E, 0.1f);
    std::vector<uint16_t> wqkv = fill((size_t)3 * E * E, 0.08f), bqkv = fill(3 * E, 0.1f), wo = fill((size_t)E * E, 0.08f), bo = fill(E, 0.1f);
    // rotary tables as the reference builds them (esme/rotary.py:110-149): inv_freq = 10000^(-2j/d), duplicated to width d, rounded to bf16
    std::vector<uint16_t> cosT((size_t)max_len * d), sinT((size_t)max_len * d);
    for (int p = 0; p < max_len; ++p)
        for (int j = 0; j < d; ++j) {
            const float inv = std::pow(10000.0f, -(float)(2 * (j % (d / 2))) / (float)d);
            cosT[(size_t)p * d + j] = f2bf(std::cos((float)p * inv));
            sinT[(size_t)p * d + j] = f2bf(std::sin((float)p * inv));
        }

    // ---------------- device side: HIP runtime for memory, the C ABI for everything else
    uint16_t *dx = upload(x), *dg = upload(gamma), *dbeta = upload(beta), *dwqkv = upload(wqkv), *dbqkv = upload(bqkv), *dwo = upload(wo), *dbo = upload(bo);
    uint16_t *dcos = upload(cosT), *dsin = upload(sinT);
    int32_t* dcu = upload(cu);
    if (!dx || !dg || !dbeta || !dwqkv || !dbqkv || !dwo || !dbo || !dcos || !dsin || !dcu) { std::fprintf(stderr, "hipMalloc / hipMemcpy failed\n"); return 2; }
    uint16_t *dh, *dqkv, *do_, *dy;
    int32_t* dpos;
    HIPCHECK(hipMalloc(&dh, (size_t)T * E * 2)); HIPCHECK(hipMalloc(&dqkv, (size_t)T * 3 * E * 2));
    HIPCHECK(hipMalloc(&do_, (size_t)T * E * 2)); HIPCHECK(hipMalloc(&dy, (size_t)T * E * 2)); HIPCHECK(hipMalloc(&dpos, (size_t)T * 4));
    hipStream_t s;
    HIPCHECK(hipStreamCreate(&s));
    ESME(esme_hip_seq_positions(dcu, B, T, dpos, nullptr, s));
    ESME(esme_hip_layernorm(dx, E, dg, dbeta, dh, E, T, E, 1e-5f, s));
    ESME(esme_hip_gemm_qkv_rotary(dh, E, dwqkv, dbqkv, dqkv, 3 * E, T, 3 * E, E, dcos, dsin, dpos, d, max_len, 2 * E, s));
    ESME(esme_hip_attn_varlen_fwd(dqkv, dqkv + E, dqkv + 2 * E, 3 * E, do_, E, dcu, B, T, H, d, max_len, 1.0f / std::sqrt((float)d), s));
    ESME(esme_hip_gemm_bf16(do_, E, dwo, dbo, dx, E, dy, E, T, E, E, ESME_EPI_RESIDUAL, 1.0f, s));
    HIPCHECK(hipStreamSynchronize(s));
    std::vector<uint16_t> y((size_t)T * E);
    HIPCHECK(hipMemcpy(y.data(), dy, y.size() * 2, hipMemcpyDeviceToHost));
    // a bad call is refused on the host with a message, before any launch
    if (esme_hip_attn_varlen_fwd(dqkv, dqkv + E, dqkv + 2 * E, 3 * E, do_, E, dcu, B, T, H, 24, max_len, 0.2f, s) == ESME_OK) { std::fprintf(stderr, "head dim 24 accepted\n"); return 4; }

    // ---------------- the same block in float64 on the host (bf16 storage at the reference's rounding points, exact arithmetic in between)
    auto r16 = [](double v) { return (double)bf2f(f2bf((float)v)); };
    std::vector<double> h((size_t)T * E), qkv((size_t)T * 3 * E), o((size_t)T * E), ref((size_t)T * E);
    for (int t = 0; t < T; ++t) {
        double mean = 0, var = 0;
        for (int e = 0; e < E; ++e) mean += bf2f(x[(size_t)t * E + e]);
        mean /= E;
        for (int e = 0; e < E; ++e) { const double c = bf2f(x[(size_t)t * E + e]) - mean; var += c * c; }
        const double rstd = 1.0 / std::sqrt(var / E + 1e-5);
        for (int e = 0; e < E; ++e) h[(size_t)t * E + e] = r16((bf2f(x[(size_t)t * E + e]) - mean) * rstd * bf2f(gamma[e]) + bf2f(beta[e]));
        for (int n = 0; n < 3 * E; ++n) {
            double acc = bf2f(bqkv[n]);
            for (int e = 0; e < E; ++e) acc += h[(size_t)t * E + e] * bf2f(wqkv[(size_t)n * E + e]);
            qkv[(size_t)t * 3 * E + n] = acc;                      // (rounded after the rotation: the GEMM epilogue rotates in fp32)
        }
    }
    for (int b = 0; b < B; ++b)
        for (int t = cu[b]; t < cu[b + 1]; ++t) {
            const int p = t - cu[b];
            for (int blk = 0; blk < 2; ++blk)                        // q and k
                for (int hh = 0; hh < H; ++hh)
                    for (int j = 0; j < d / 2; ++j) {
                        double& lo = qkv[(size_t)t * 3 * E + blk * E + hh * d + j];
                        double& up = qkv[(size_t)t * 3 * E + blk * E + hh * d + j + d / 2];
                        const double c0 = bf2f(cosT[(size_t)p * d + j]), s0 = bf2f(sinT[(size_t)p * d + j]);
                        const double nl = lo * c0 - up * s0, nu = up * c0 + lo * s0;
                        lo = nl; up = nu;
                    }
            for (int n = 0; n < 3 * E; ++n) qkv[(size_t)t * 3 * E + n] = r16(qkv[(size_t)t * 3 * E + n]);
        }
    for (int b = 0; b < B; ++b)
        for (int hh = 0; hh < H; ++hh)
            for (int t = cu[b]; t < cu[b + 1]; ++t) {
                std::vector<double> sc(cu[b + 1] - cu[b]);
                double mx = -1e300, sum = 0;
                for (int u = cu[b]; u < cu[b + 1]; ++u) {
                    double a = 0;
                    for (int j = 0; j < d; ++j) a += qkv[(size_t)t * 3 * E + hh * d + j] * qkv[(size_t)u * 3 * E + E + hh * d + j];
                    sc[u - cu[b]] = a / std::sqrt((double)d);
                    mx = sc[u - cu[b]] > mx ? sc[u - cu[b]] : mx;
                }
                for (auto& v : sc) { v = std::exp(v - mx); sum += v; }
                for (int j = 0; j < d; ++j) {
                    double a = 0;
                    for (int u = cu[b]; u < cu[b + 1]; ++u) a += sc[u - cu[b]] * qkv[(size_t)u * 3 * E + 2 * E + hh * d + j];
                    o[(size_t)t * E + hh * d + j] = r16(a / sum);
                }
            }
    double num = 0, den = 0, worst = 0;
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < E; ++n) {
            double acc = bf2f(bo[n]);
            for (int e = 0; e < E; ++e) acc += o[(size_t)t * E + e] * bf2f(wo[(size_t)n * E + e]);
            const double r = bf2f(x[(size_t)t * E + n]) + acc, g = bf2f(y[(size_t)t * E + n]);
            num += (g - r) * (g - r); den += r * r;
            worst = std::fabs(g - r) > worst ? std::fabs(g - r) : worst;
        }
    const double rel = std::sqrt(num / den);
    std::printf("c_abi_demo: T = %d residues in %d sequences, E = %d: rel-Frobenius %.3e, max |diff| %.3e vs the float64 host block\n", T, B, E, rel, worst);
    // bf16 storage at four points (LayerNorm, q/k/v, attention output with a bf16 P, the stream): 2^-7 is the per-kernel bar of tests/test_hip_kernels.py
    if (!(rel < 7.8125e-3)) { std::fprintf(stderr, "MISMATCH\n"); return 5; }
    std::printf("c_abi_demo: OK\n");
    return 0;
}
