// HBM-bound row kernels of the packed forward pass: embedding gather, sequence
// positions, LayerNorm, varlen rotary, row softmax, row gather/scatter.
// All of them move 16 B per lane per access along the packed-residue axis; none of
// them has inter-block reuse, so no XCD remap (guide T1: 0 % on LayerNorm).
#include "common.h"
#include "launch.h"

namespace esme {

// ---------------------------------------------------------------- embedding
// one lane per 16-byte chunk of an output row
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ tokens,
                                                    const u32x4* __restrict__ table, u32x4* __restrict__ out,
                                                    int64_t T, int chunks, int V, int mask_idx, int pad_idx) {
    const int64_t total = T * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / chunks;
        const int c = (int)(i - t * chunks);
        const int64_t tok = tokens[t];
        u32x4 v = {0u, 0u, 0u, 0u};
        if (tok != mask_idx && tok != pad_idx && tok >= 0 && tok < V) v = table[tok * chunks + c];
        out[i] = v;
    }
}

// token row (+ `<mask>` zeroing) + learned-position row, one rounding: ESM-1b / ESM-1v
__global__ __launch_bounds__(256) void embed_pos_kernel(const int64_t* __restrict__ tokens,
                                                        const u32x4* __restrict__ table,
                                                        const u32x4* __restrict__ pos_table,
                                                        const int32_t* __restrict__ pos_idx, int pos_offset,
                                                        u32x4* __restrict__ out, float* __restrict__ out32, int64_t T, int chunks, int V,
                                                        int P, int mask_idx) {
    const int64_t total = T * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / chunks;
        const int c = (int)(i - t * chunks);
        const int64_t tok = tokens[t];
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b[8];
        if (tok != mask_idx && tok >= 0 && tok < V) unpack8(table[tok * chunks + c], a);
        int p = pos_idx[t] + pos_offset;
        p = p < 0 ? 0 : (p < P ? p : P - 1);
        unpack8(pos_table[(int64_t)p * chunks + c], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        if (out32) {                          // split-operand ('exact') mode: the sum of two bf16 values, exactly, in fp32
            float* o = out32 + i * 8;
            *reinterpret_cast<f32x4*>(o) = f32x4{a[0], a[1], a[2], a[3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{a[4], a[5], a[6], a[7]};
        } else {
            out[i] = pack8(a);
        }
    }
}

// ------------------------------------------------------- sequence positions
// binary search of the row in cu_lens (B+1 entries, L2 resident)
__global__ __launch_bounds__(256) void seqpos_kernel(const int32_t* __restrict__ cu, int B, int64_t T,
                                                     int32_t* __restrict__ pos, int32_t* __restrict__ seq) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    int lo = 0, hi = B;              // invariant: cu[lo] <= t < cu[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)cu[mid] <= t) lo = mid; else hi = mid;
    }
    if (pos) pos[t] = (int32_t)(t - cu[lo]);
    if (seq) seq[t] = lo;
}

// order[rank] = i with rank = #{j : len_j > len_i or (len_j == len_i and j < i)}: longest sequence first, stable.  One
// workgroup, lengths in LDS, B <= 1024 (each thread ranks one sequence against all others: <= 1024 LDS reads).
__global__ __launch_bounds__(1024) void seq_order_kernel(const int32_t* __restrict__ cu, int B, int32_t* __restrict__ order) {
    __shared__ int len[1024];
    const int i = threadIdx.x;
    if (B > 1024) {                       // too many for the rank sort: identity
        for (int k = i; k < B; k += 1024) order[k] = k;
        return;
    }
    if (i < B) len[i] = cu[i + 1] - cu[i];
    __syncthreads();
    if (i >= B) return;
    const int li = len[i];
    int rank = 0;
    for (int j = 0; j < B; ++j) {
        const int lj = len[j];
        rank += (lj > li) || (lj == li && j < i);
    }
    order[rank] = i;
}

// ---------------------------------------------------------------- LayerNorm
// One wave per row; the row stays in registers (NCH chunks of 8 bf16 per lane), so HBM
// traffic is exactly one read + one write of the row: 4*E bytes.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const u16* __restrict__ x, int64_t ldx,
                                                        const u16* __restrict__ w, const u16* __restrict__ b,
                                                        u16* __restrict__ y, int64_t ldy, int64_t T, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const u16* xr = x + row * ldx;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
            unpack8(*reinterpret_cast<const u32x4*>(xr + e0), v[c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    const float inv_e = 1.0f / (float)E;
    const float mean = wave_sum(s) * inv_e;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; ss = fmaf(d, d, ss); }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) * inv_e + eps);
    // (every product and sum below is spelled out -- no contraction freedom -- so that this kernel and the fused q/k LayerNorm + rotary pass, which
    // must equal it bit for bit, compute the same bits in every instantiation: hipcc's own fma pairing differed between NCH = 2 / 10 and the rest)
    u16* yr = y + row * ldy;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
            float wf[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(w + e0), wf);
            if (b) {
                float bfv[8];
                unpack8(*reinterpret_cast<const u32x4*>(b + e0), bfv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaf(__fmul_rn(v[c][j] - mean, rstd), wf[j], bfv[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = __fmul_rn(__fmul_rn(v[c][j] - mean, rstd), wf[j]);
            }
            *reinterpret_cast<u32x4*>(yr + e0) = pack8(o);
        }
    }
}


// ------------------------------------------- high-precision mode: fp32 residual stream
// x32 <- (init ? 0 : x32) + alpha * o   (o: a branch output in bf16; x32: the fp32 residual stream), plus what the next
// LayerNorm-folded GEMM needs: x16 = bf16(x32) (its MFMA operand) and the row's {sum, sum of squares} of the fp32 values.
// One wave per row, HBM-bound: 2 (o) + 4 + 4 (x32 r/w) + 2 (x16) bytes per element.
template <int NCH>
__global__ __launch_bounds__(256) void residual_f32_kernel(float* __restrict__ x32, int64_t ld32, const u16* __restrict__ o,
                                                           int64_t ldo, float alpha, int init, u16* __restrict__ x16,
                                                           int64_t ld16, f32x2* __restrict__ sums, int64_t T, int E) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    float* xr = x32 + row * ld32;
    const u16* orow = o + row * ldo;
    u16* yr = x16 + row * ld16;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
            float v[8], b[8];
            unpack8(*reinterpret_cast<const u32x4*>(orow + e0), b);
            if (init) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = alpha * b[j];
            } else {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(xr + e0), a1 = *reinterpret_cast<const f32x4*>(xr + e0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = fmaf(alpha, b[j], a0[j]); v[4 + j] = fmaf(alpha, b[4 + j], a1[j]); }
            }
            *reinterpret_cast<f32x4*>(xr + e0) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(xr + e0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
            *reinterpret_cast<u32x4*>(yr + e0) = pack8(v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0 && sums) sums[row] = f32x2{s1, s2};
}

// LayerNorm of an fp32 row into bf16 (the final LayerNorm of the high-precision mode)
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, int64_t ldx, const u16* __restrict__ w,
                                                            const u16* __restrict__ b, u16* __restrict__ y, int64_t ldy,
                                                            int64_t T, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float* xr = x + row * ldx;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(xr + e0), a1 = *reinterpret_cast<const f32x4*>(xr + e0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[c][j] = a0[j]; v[c][4 + j] = a1[j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    const float inv_e = 1.0f / (float)E;
    const float mean = wave_sum(s) * inv_e;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; ss += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) * inv_e + eps);
    u16* yr = y + row * ldy;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
            float wf[8], o[8], bfv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            unpack8(*reinterpret_cast<const u32x4*>(w + e0), wf);
            if (b) unpack8(*reinterpret_cast<const u32x4*>(b + e0), bfv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd * wf[j] + bfv[j];
            *reinterpret_cast<u32x4*>(yr + e0) = pack8(o);
        }
    }
}

// ------------------------------------------- split-operand ('exact') mode: LayerNorm into a (hi, lo) bf16 pair
// y = LayerNorm(x) in fp32, written as hi = bf16(y), lo = bf16(y - hi) (lo at column out_off + e of the same row): the MFMA
// operand pair of the next K-doubled GEMM, optionally also as fp32 (y32: the representation the model
// returns).  Input: the fp32 residual stream (IN = 0) or a (hi, lo) pair read as hi + lo (IN = 1: the LM head's LayerNorm).
// One wave per row, two-pass fp32 statistics with the row in registers; 4 + 4 (+ 4) bytes per element.
template <int NCH, int IN>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const void* __restrict__ xv, int64_t ldx, int64_t in_off,
                                                              const u16* __restrict__ w, const u16* __restrict__ b,
                                                              u16* __restrict__ y, int64_t ldy, int64_t out_off,
                                                              float* __restrict__ y32, int64_t ld32, int64_t T, int E, float eps,
                                                              int* __restrict__ ovf) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
            if constexpr (IN == 0) {
                const float* xr = reinterpret_cast<const float*>(xv) + row * ldx;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(xr + e0), a1 = *reinterpret_cast<const f32x4*>(xr + e0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[c][j] = a0[j]; v[c][4 + j] = a1[j]; }
            } else {
                const u16* xr = reinterpret_cast<const u16*>(xv) + row * ldx;
                float lo[8];
                unpack8t<IN == 2>(*reinterpret_cast<const u32x4*>(xr + e0), v[c]);          // (IN == 2: an fp16 pair, precision 'half')
                unpack8t<IN == 2>(*reinterpret_cast<const u32x4*>(xr + in_off + e0), lo);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] += lo[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    const float inv_e = 1.0f / (float)E;
    const float mean = wave_sum(s) * inv_e;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; ss += d * d; }
        }
    }
    const float var_e = wave_sum(ss);
    // range guard of precision 'half': a stream value past fp16's 65 504 arrives here as inf / NaN (esme_gemm_fusion_t.overflow_flag)
    if (ovf && lane == 0 && !(var_e < 3.0e38f)) atomicOr(ovf, 1);
    const float rstd = 1.0f / sqrtf(var_e * inv_e + eps);       // (correctly rounded forms: this mode is about accuracy)
    u16* yr = y + row * ldy;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
            float wf[8], o[8], bfv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, hi[8];
            unpack8(*reinterpret_cast<const u32x4*>(w + e0), wf);
            if (b) unpack8(*reinterpret_cast<const u32x4*>(b + e0), bfv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd * wf[j] + bfv[j];
            const u32x4 ph = pack8(o);
            unpack8(ph, hi);
            *reinterpret_cast<u32x4*>(yr + e0) = ph;
#pragma unroll
            for (int j = 0; j < 8; ++j) hi[j] = o[j] - hi[j];
            *reinterpret_cast<u32x4*>(yr + out_off + e0) = pack8(hi);
            if (y32) {
                float* zr = y32 + row * ld32 + e0;
                *reinterpret_cast<f32x4*>(zr) = f32x4{o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(zr + 4) = f32x4{o[4], o[5], o[6], o[7]};
            }
        }
    }
}

// ------------------------------------------------------- LayerNorm statistics
// sums[row] = {sum x, sum x^2}: one wave per row (the first layer's input; later layers get
// their statistics from the residual GEMM epilogues)
template <int NCH>
__global__ __launch_bounds__(256) void row_sums_kernel(const u16* __restrict__ x, int64_t ldx, int64_t T, int E,
                                                       f32x2* __restrict__ sums) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const u16* xr = x + row * ldx;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(xr + e0), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1 += v[j]; s2 += v[j] * v[j]; }
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) sums[row] = f32x2{s1, s2};
}

// ------------------------------------------------------------------- rotary
// One lane per (row, tensor, head, 8-wide chunk of the first half of the head): it
// rotates that chunk together with its partner chunk d/2 further on.  Tables are tiny
// (max_len*d*2 B) and L2 resident; q and k rows are read and written once: 8*E B/row.
template <bool F16>          // (true: q, k and the tables are IEEE fp16 -- precision 'half' at head dims the QKV epilogue does not rotate)
__global__ __launch_bounds__(256) void rotary_kernel(u16* __restrict__ q, u16* __restrict__ k, int64_t ld,
                                                     const u16* __restrict__ cosT, const u16* __restrict__ sinT,
                                                     const int32_t* __restrict__ pos, int64_t T, int H, int d,
                                                     int max_len) {
    const int half_chunks = d >> 4;                 // 8-wide chunks in d/2
    const int per_tensor = H * half_chunks;         // work items per row per tensor
    const int per_row = 2 * per_tensor;
    const int64_t total = T * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / per_row;
        int r = (int)(i - t * per_row);
        u16* base = q;
        if (r >= per_tensor) { r -= per_tensor; base = k; }
        const int h = r / half_chunks;
        const int jc = r - h * half_chunks;
        int p = pos[t];
        p = p < max_len ? p : max_len - 1;
        u16* xp = base + t * ld + h * d + jc * 8;
        float lo[8], hi[8], c[8], s[8], olo[8], ohi[8];
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(xp), lo);
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(xp + (d >> 1)), hi);
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(cosT + (int64_t)p * d + jc * 8), c);
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(sinT + (int64_t)p * d + jc * 8), s);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // one product rounded, one fused, spelled out: the same contraction in every kernel that rotates
            olo[j] = fmaf(lo[j], c[j], -__fmul_rn(hi[j], s[j]));
            ohi[j] = fmaf(hi[j], c[j], __fmul_rn(lo[j], s[j]));
        }
        *reinterpret_cast<u32x4*>(xp) = pack8t<F16>(olo);
        *reinterpret_cast<u32x4*>(xp + (d >> 1)) = pack8t<F16>(ohi);
    }
}


// ------------------------------------------- split-operand ('exact') mode: rotary on (hi, lo) pairs with fp32 tables
// The reference's fp32 forward rotates with fp32 cos / sin (esme/rotary.py:144-149 casts the tables to the activation dtype), so
// this mode cannot use the bf16 tables of the fused QKV epilogue (their rounding alone costs 5e-4 on the logits).  In place on
// `nheads` consecutive heads of width d (the q and k blocks of the projection's pair output): x = hi + lo in fp32,
// x[j] <- x[j] cos - x[j + d/2] sin, x[j + d/2] <- x[j + d/2] cos + x[j] sin, written back as a pair.  One lane per 8-wide chunk
// pair (j, j + d/2); HBM-bound: 16 B per element (hi and lo, read and written).
template <bool F16>
__global__ __launch_bounds__(256) void rotary_split_kernel(u16* __restrict__ x, int64_t ld, int64_t lo_off,
                                                           const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                           const int32_t* __restrict__ pos, int64_t T, int nheads, int d, int max_len) {
    const int cph = d >> 4;                               // chunk pairs per head
    const int64_t total = T * nheads * cph;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int c = (int)(it % cph);
        const int64_t th = it / cph;
        const int h = (int)(th % nheads);
        const int64_t t = th / nheads;
        int p = pos[t];
        p = p < max_len ? p : max_len - 1;
        u16* xp = x + t * ld + h * d + c * 8;
        float lh[8], ll[8], uh[8], ul[8], cs[8], sn[8];
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(xp), lh);
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(xp + lo_off), ll);
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(xp + (d >> 1)), uh);
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(xp + (d >> 1) + lo_off), ul);
        const float* cp = cosT + (int64_t)p * d + c * 8;
        const float* sp = sinT + (int64_t)p * d + c * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { cs[j] = c0[j]; cs[4 + j] = c1[j]; sn[j] = s0[j]; sn[4 + j] = s1[j]; }
        float olo[8], oup[8], rl[8], ru[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float lo = lh[j] + ll[j], up = uh[j] + ul[j];
            olo[j] = fmaf(lo, cs[j], -__fmul_rn(up, sn[j]));
            oup[j] = fmaf(up, cs[j], __fmul_rn(lo, sn[j]));
        }
        const u32x4 plo = pack8t<F16>(olo), pup = pack8t<F16>(oup);
        unpack8t<F16>(plo, rl);
        unpack8t<F16>(pup, ru);
#pragma unroll
        for (int j = 0; j < 8; ++j) { rl[j] = olo[j] - rl[j]; ru[j] = oup[j] - ru[j]; }
        *reinterpret_cast<u32x4*>(xp) = plo;
        *reinterpret_cast<u32x4*>(xp + lo_off) = pack8t<F16>(rl);
        *reinterpret_cast<u32x4*>(xp + (d >> 1)) = pup;
        *reinterpret_cast<u32x4*>(xp + (d >> 1) + lo_off) = pack8t<F16>(ru);
    }
}

// ------------------------------------------- MFMA operand of an fp32 stream
// x16 = round(x32) (bf16, or IEEE fp16 for precision 'half') + per-row {sum, sum of squares} of the ROUNDED values: the operand the
// LayerNorm-folded GEMMs read and the statistics they fold (the start of a forward on an fp32 residual stream whose rows are not
// bf16 embedding rows, e.g. after the learned-position sum; afterwards the residual GEMMs' epilogues keep both current).
template <int NCH, bool F16>
__global__ __launch_bounds__(256) void stream_operand_kernel(const float* __restrict__ x32, int64_t ld32, u16* __restrict__ x16,
                                                             int64_t ld16, int64_t lo_off, const float* __restrict__ scale,
                                                             const int32_t* __restrict__ ext_sel, int ext_n, int64_t ext_off,
                                                             f32x2* __restrict__ sums, unsigned int* __restrict__ col_absmax, int64_t T, int E) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float* xr = x32 + row * ld32;
    u16* yr = x16 + row * ld16;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        if (e0 < E) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(xr + e0), a1 = *reinterpret_cast<const f32x4*>(xr + e0 + 4);
            float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            if (lo_off) {                                  // pair stream: statistics of the fp32 value itself (what the residual epilogues emit later)
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
                if (scale) {                               // stored = rho * x (esme_gemm_fusion_t.pair_scale_in / _out)
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(scale + e0), c1 = *reinterpret_cast<const f32x4*>(scale + e0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = __fmul_rn(v[j], c0[j]); v[4 + j] = __fmul_rn(v[4 + j], c1[j]); }
                }
                if (col_absmax) {
                    // plan guard of precision 'half' (esme_hip_stream_operand_guarded): running max |stored value| per column.  The current maxima are READ first
                    // (plain loads, possibly stale -- they only decide whether an atomic is worth issuing): a running maximum is raised O(log T) times per
                    // column, so after the first few rows almost no lane issues one.
                    const u32x4 m0 = *reinterpret_cast<const u32x4*>(col_absmax + e0), m1 = *reinterpret_cast<const u32x4*>(col_absmax + e0 + 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned int b = __float_as_uint(fabsf(v[j]));
                        if (b > (j < 4 ? m0[j] : m1[j - 4])) atomicMax(col_absmax + e0 + j, b);
                    }
                }
            }
            const u32x4 pk = pack8t<F16>(v);
            *reinterpret_cast<u32x4*>(yr + e0) = pk;
            float r[8];
            unpack8t<F16>(pk, r);
            if (lo_off) {                                  // lo = round(x - hi), lo_off columns further in the same row
                float l[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) l[j] = v[j] - r[j];
                *reinterpret_cast<u32x4*>(yr + lo_off + e0) = pack8t<F16>(l);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1 += r[j]; s2 = fmaf(r[j], r[j], s2); }
            }
        }
    }
    if (ext_off) {                                         // extension K-tile of the pair stream: lo of the selected channels, zeros behind them
        u16 val = 0;
        if (lane < ext_n) {
            int c = ext_sel[lane];                          // (a device-side list: clamped, so a bad entry cannot read outside the row)
            c = c < 0 ? 0 : (c < E ? c : E - 1);
            const float v = scale ? __fmul_rn(xr[c], scale[c]) : xr[c];
            val = f2h<F16>(v - h2f<F16>(f2h<F16>(v)));
        }
        yr[ext_off + lane] = val;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0 && sums) sums[row] = f32x2{s1, s2};
}

// out32 = hi + lo of a 16-bit pair stream (bf16 or fp16): the raw layer outputs `forward_representation(layers=[...])` returns in the pair modes
template <bool F16>
__global__ __launch_bounds__(256) void pair_to_f32_kernel(const u16* __restrict__ x, int64_t ld, int64_t lo_off, float* __restrict__ out,
                                                          int64_t ld32, int64_t T, int chunks) {
    const int64_t total = T * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / chunks;
        const int c = (int)(i - t * chunks);
        float h[8], l[8];
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(x + t * ld + c * 8), h);
        unpack8t<F16>(*reinterpret_cast<const u32x4*>(x + t * ld + lo_off + c * 8), l);
        float* o = out + t * ld32 + c * 8;
        *reinterpret_cast<f32x4*>(o) = f32x4{h[0] + l[0], h[1] + l[1], h[2] + l[2], h[3] + l[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{h[4] + l[4], h[5] + l[5], h[6] + l[6], h[7] + l[7]};
    }
}

// ------------------------------------------- q/k LayerNorm + rotary (ESM-C)
// ESM-C normalises q and k over the FULL embedding width between the projection and the rotary
// (attention.py:104-105), so that LayerNorm cannot ride in a GEMM epilogue (a row spans several
// column tiles).  A wave takes (q | k) of RPW consecutive rows: a row stays in registers, gets normalised,
// rounded to bf16 (the value the stand-alone LayerNorm kernel would have written) and rotated in place; the
// rotary partner of a lane's 8 elements (d/2 further inside the head) lives d/16 lanes away, so
// the exchange is four 32-bit lane shuffles.  One read + one write of q and k: 8*E bytes per row,
// instead of three passes (two LayerNorms + rotary).
// The LayerNorm weights (biases) of the wave's type are loaded once for its RPW rows, and the cos / sin chunk of a lane is the same for
// every 64-lane chunk of the row (512 elements per chunk step is a multiple of the head dim), so it is ONE load each per row: 8 + 6 / RPW
// vector-memory instructions per 2.3 KB row item.  What bounds the pass since then is LATENCY x occupancy (round 5, profiles/r05_qk_norm_ab.txt):
// a row's first store waits for a full-row reduction and the kernel needs ~100 - 120 VGPRs (4 waves per SIMD), so it is written memory-first
// (positions of all the wave's rows, then the next row and its table chunks requested before the current row is reduced) and VALU-lean.
// F16 (precision 'half'): q, k and the rotary tables are IEEE fp16 (the LayerNorm parameters stay bf16).
// Sum over the 64 lanes, every lane gets the total: the xor-butterfly 32, 16, 8, 4, 2, 1 of wave_sum() -- the SAME additions in the same order, hence the same
// bits -- on the VALU's cross-lane paths (v_permlane32_swap / v_permlane16_swap, DPP row_ror / quad_perm) instead of six dependent ds_bpermute round trips
// through the LDS unit.  (After a step, lanes that are congruent modulo the step hold identical sums, so a rotation by the step is as good as the xor.)
__device__ __forceinline__ float wave_sum_valu(float v) {
    {
        const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);      // {[lo | lo], [hi | hi]}: own + lane ^ 32
        v = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    }
    {
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);      // own + lane ^ 16
        v = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    }
    v += dpp_f32<0x128>(v);      // row_ror:8
    v += dpp_f32<0x124>(v);      // row_ror:4
    v += dpp_f32<0x4E>(v);       // quad_perm [2, 3, 0, 1]: lane ^ 2
    v += dpp_f32<0xB1>(v);       // quad_perm [1, 0, 3, 2]: lane ^ 1
    return v;
}

template <int NCH, int RPW, bool F16 = false>
__global__ __launch_bounds__(256) void qk_norm_rotary_kernel(u16* __restrict__ q, u16* __restrict__ k, int64_t ld,
                                                             const u16* __restrict__ wq, const u16* __restrict__ wk,
                                                             const u16* __restrict__ bq, const u16* __restrict__ bk,
                                                             float eps, const u16* __restrict__ cosT,
                                                             const u16* __restrict__ sinT, const int32_t* __restrict__ pos,
                                                             int64_t T, int E, int d, int max_len, float q_scale, unsigned int* __restrict__ qk_sumsq) {
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;                     // waves 0, 1: q; 2, 3: k
    float gmax[NCH];                                     // plan guard (F16, qk_sumsq != NULL): running max over this wave's rows of the squared row norm of the head this lane sits in, per chunk
#pragma unroll
    for (int c = 0; c < NCH; ++c) gmax[c] = 0.f;
    const bool is_k = wv >= 2;
    const int64_t row0 = ((int64_t)blockIdx.x * 2 + (wv & 1)) * RPW;
    if (row0 >= T) return;
    const u16* w = is_k ? wk : wq;
    const u16* b = is_k ? bk : bq;
    // The pass is bound by VALU ISSUE, not by bytes (round 5: ~600 vector instructions per 2.3 KB row item = 63 us of issue per SIMD at the ESMC-600M shape,
    // against 47 us of memory time): the LayerNorm weights of the wave's rows are unpacked once, the rotation's sign is folded into the sine once per row,
    // lanes past the row's end skip the arithmetic (the rotary partner lane ^ d/16 of an active lane is active: E is a multiple of d), and the two row
    // reductions run on the VALU's cross-lane paths.
    float wf[NCH][8];
    u32x4 braw[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e0 = (c * 64 + lane) * 8;
        braw[c] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[c][j] = 0.f;
        if (e0 < E) {
            unpack8(*reinterpret_cast<const u32x4*>(w + e0), wf[c]);
            if (b) braw[c] = *reinterpret_cast<const u32x4*>(b + e0);
        }
    }
    const int half = d >> 1, shift = d >> 4;            // partner lane = lane ^ shift
    const int local = (lane * 8) % d;                   // position inside the head: the same for every chunk step
    const bool lower = local < half;
    const int jc = lower ? local : local - half;
    const float sgn = lower ? -1.0f : 1.0f;             // lower half: lo c - up s; upper: up c + lo s
    const float inv_e = 1.0f / (float)E;
    // Memory first: a row's first store waits for a full-row reduction, so what hides the load latency is rows in flight.  The positions of all the wave's
    // rows and the first row are requested before anything else; inside the loop the NEXT row and its table chunks are requested before this row is reduced.
    int prow[RPW];
#pragma unroll
    for (int it = 0; it < RPW; ++it) prow[it] = pos[row0 + it < T ? row0 + it : T - 1];
    u32x4 cur[NCH], nxt[NCH];
    {
        const u16* x0 = (is_k ? k : q) + row0 * ld;
#pragma unroll
        for (int c = 0; c < NCH; ++c) { const int e0 = (c * 64 + lane) * 8; cur[c] = e0 < E ? *reinterpret_cast<const u32x4*>(x0 + e0) : u32x4{0u, 0u, 0u, 0u}; }
    }
    u32x4 craw, sraw, cnxt, snxt;
    {
        int p = prow[0];
        p = p < max_len ? p : max_len - 1;
        craw = *reinterpret_cast<const u32x4*>(cosT + (int64_t)p * d + jc);
        sraw = *reinterpret_cast<const u32x4*>(sinT + (int64_t)p * d + jc);
    }
#pragma unroll
    for (int it = 0; it < RPW; ++it) {
        const int64_t row = row0 + it;
        if (row >= T) break;
        u16* xr = (is_k ? k : q) + row * ld;
        if (it + 1 < RPW && row + 1 < T) {
            const u16* xn = xr + ld;
#pragma unroll
            for (int c = 0; c < NCH; ++c) { const int e0 = (c * 64 + lane) * 8; nxt[c] = e0 < E ? *reinterpret_cast<const u32x4*>(xn + e0) : u32x4{0u, 0u, 0u, 0u}; }
            int p = prow[it + 1 < RPW ? it + 1 : it];
            p = p < max_len ? p : max_len - 1;
            cnxt = *reinterpret_cast<const u32x4*>(cosT + (int64_t)p * d + jc);
            snxt = *reinterpret_cast<const u32x4*>(sinT + (int64_t)p * d + jc);
        }
        float v[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int e0 = (c * 64 + lane) * 8;
            if (e0 < E) {
                unpack8t<F16>(cur[c], v[c]);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[c][j];
            }
        }
        const float mean = wave_sum_valu(s) * inv_e;
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int e0 = (c * 64 + lane) * 8;
            if (e0 < E) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float dv = v[c][j] - mean; ss = fmaf(dv, dv, ss); }
            }
        }
        const float rstd = rsqrtf(wave_sum_valu(ss) * inv_e + eps);
        float cs[8], sn[8];
        unpack8t<F16>(craw, cs);
        unpack8t<F16>(sraw, sn);
#pragma unroll
        for (int j = 0; j < 8; ++j) sn[j] *= sgn;            // (exact: fmul(o2, -s) = -fmul(o2, s))
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int e0 = (c * 64 + lane) * 8;
            if (e0 < E) {                                     // (lane-divergent only in the last chunk; the partner lane takes the same branch)
                float o[8];
                // (every product and sum spelled out: no contraction freedom, the same bits in every instantiation)
                if (b) {
                    float bfv[8];
                    unpack8(braw[c], bfv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = fmaf(__fmul_rn(v[c][j] - mean, rstd), wf[c][j], bfv[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = __fmul_rn(__fmul_rn(v[c][j] - mean, rstd), wf[c][j]);
                }
                float a[8], o2[8];
                if constexpr (F16) {                          // precision 'half' answers to the fp32 forward and keeps the fp32 values: one rounding fewer on q, k
#pragma unroll
                    for (int j = 0; j < 8; ++j) { a[j] = o[j]; o2[j] = __shfl_xor(o[j], shift, 64); }
                } else {
                    const u32x4 y = pack8(o);                 // bf16 rounding point of the LayerNorm output (the reference's)
                    u32x4 other;
#pragma unroll
                    for (int i = 0; i < 4; ++i) other[i] = (unsigned int)__shfl_xor((int)y[i], shift, 64);
                    unpack8(y, a);
                    unpack8(other, o2);
                }
                float r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = fmaf(a[j], cs[j], __fmul_rn(o2[j], sn[j]));
                if constexpr (F16) {
                    if (qk_sumsq) {                           // (wave-uniform) |r|^2 over the head: the rotation preserves it; lanes of a head are neighbours
                        float ss = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) ss = fmaf(r[j], r[j], ss);
                        if (d >= 16) ss += dpp_f32<0xB1>(ss);
                        if (d >= 32) ss += dpp_f32<0x4E>(ss);
                        if (d >= 64) ss += dpp_f32<0x141>(ss);
                        if (d >= 128) ss += dpp_f32<0x128>(ss);
                        gmax[c] = fmaxf(gmax[c], ss);
                    }
                }
                if (!is_k && q_scale != 1.0f) {              // softmax_scale * log2(e) folded into q (fp32, before the rounding): attention's q_prescaled
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[j] *= q_scale;
                }
                *reinterpret_cast<u32x4*>(xr + e0) = pack8t<F16>(r);
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
        craw = cnxt; sraw = snxt;
    }
    if constexpr (F16) {
        if (qk_sumsq) {                                       // one filtered atomic per head and wave (esme_gemm_fusion_t.qk_sumsq's layout: [q | k][head])
            const int lph = d >> 3;                           // lanes per head
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e0 = (c * 64 + lane) * 8;
                if (e0 < E && (lane % lph) == 0) {
                    unsigned int* slot = qk_sumsq + (is_k ? E / d : 0) + e0 / d;
                    const unsigned int b = __float_as_uint(gmax[c]);
                    if (b > *slot) atomicMax(slot, b);
                }
            }
        }
    }
}

// ------------------------------------------------------------- row softmax
// V <= 64: one wave per row, one lane per column.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const u16* __restrict__ x, int64_t ldx,
                                                           u16* __restrict__ y, int64_t ldy, int64_t T, int V,
                                                           int log_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float v = lane < V ? bf2f(x[row * ldx + lane]) : -INFINITY;
    const float m = wave_max(v);
    const float e = lane < V ? __expf(v - m) : 0.f;
    const float sum = wave_sum(e);
    if (lane < V) y[row * ldy + lane] = f2bf(log_flag ? (v - m) - __logf(sum) : e / sum);
}

// fp32 in / fp32 out (the split-operand mode's logits)
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(const float* __restrict__ x, int64_t ldx,
                                                               float* __restrict__ y, int64_t ldy, int64_t T, int V,
                                                               int log_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float v = lane < V ? x[row * ldx + lane] : -INFINITY;
    const float m = wave_max(v);
    const float e = lane < V ? expf(v - m) : 0.f;
    const float sum = wave_sum(e);
    if (lane < V) y[row * ldy + lane] = log_flag ? (v - m) - logf(sum) : e / sum;
}

// ------------------------------------------------------ row gather / scatter
__global__ __launch_bounds__(256) void gather_rows_kernel(const u32x4* __restrict__ src,
                                                          const int64_t* __restrict__ idx, u32x4* __restrict__ dst,
                                                          int64_t n, int chunks, int scatter, int64_t bound) {
    // `bound` = rows of the INDEXED side (src for a gather, dst for a scatter).  An index outside
    // [0, bound) never touches memory: a scatter drops the row, a gather delivers zeros (the
    // reference's torch indexing raises there; a raw kernel must at least not corrupt HBM).
    const int64_t total = n * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks);
        const int64_t j = idx[r];
        const bool ok = j >= 0 && j < bound;
        if (scatter) { if (ok) dst[j * chunks + c] = src[i]; }
        else dst[i] = ok ? src[j * chunks + c] : u32x4{0u, 0u, 0u, 0u};
    }
}


// ------------------------------------------------------- per-sequence mean
// One workgroup per (sequence, 64-column slab): 8 lanes cover the slab's 64 columns (8 per lane, whole
// 128-B lines of bf16), the 32 lane groups stride the sequence's rows, fp32 accumulation, then a
// fixed-order LDS reduction (deterministic).  B * E/64 workgroups keep the chip busy even for a few long
// proteins (the first version used 512-column slabs: 96 workgroups for 32 proteins, 1.1 TB/s).
template <bool F32>
__global__ __launch_bounds__(256) void segment_mean_kernel(const void* __restrict__ xv, int64_t ldx,
                                                           const int32_t* __restrict__ cu, int E,
                                                           void* __restrict__ outv, int64_t ldo) {
    __shared__ float part[32][64 + 4];
    const int cg = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const int seq = blockIdx.x;
    const int col = blockIdx.y * 64 + cg * 8;
    const int a = cu[seq], b = cu[seq + 1];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < E) {
        for (int r = a + rg; r < b; r += 32) {
            float f[8];
            if (F32) {
                const float* x = (const float*)xv + (int64_t)r * ldx + col;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(x);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(x + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
            } else {
                unpack8(*reinterpret_cast<const u32x4*>((const u16*)xv + (int64_t)r * ldx + col), f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[rg][cg * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.y * 64 + threadIdx.x < E) {
        const int c = threadIdx.x;
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) s += part[g][c];
        s *= b > a ? 1.0f / (float)(b - a) : 0.f;
        const int64_t o = (int64_t)seq * ldo + blockIdx.y * 64 + c;
        if (F32) ((float*)outv)[o] = s;
        else ((u16*)outv)[o] = f2bf(s);
    }
}

}  // namespace esme

using namespace esme;

static inline unsigned int grid_for(int64_t items, int per_block, unsigned int cap = 256u * 16u) {
    int64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (unsigned int)(g > (int64_t)cap ? cap : g);
}

extern "C" int esme_hip_embed(const int64_t* tokens, const void* table, void* out, int64_t T, int E, int V,
                              int mask_idx, int pad_idx, void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0 && V > 0, "embed: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(tokens && table && out, "embed: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && aligned16(table) && aligned16(out), "embed: E %% 8 != 0 or misaligned");
    const int chunks = E / 8;
    hipLaunchKernelGGL(embed_kernel, dim3(grid_for(T * chunks, 256)), dim3(256), 0, (hipStream_t)stream, tokens,
                       (const u32x4*)table, (u32x4*)out, T, chunks, V, mask_idx, pad_idx);
    return check_launch("embed");
}

extern "C" int esme_hip_seq_positions(const int32_t* cu_lens, int B, int64_t T, int32_t* pos, int32_t* seq_id,
                                      void* stream) {
    ESME_CHECK_ARG(B >= 0 && T >= 0, "seq_positions: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(cu_lens && B >= 1, "seq_positions: null cu_lens or B < 1 with T > 0");
    hipLaunchKernelGGL(seqpos_kernel, dim3((unsigned int)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       cu_lens, B, T, pos, seq_id);
    return check_launch("seq_positions");
}

extern "C" int esme_hip_seq_order(const int32_t* cu_lens, int B, int32_t* order, void* stream) {
    ESME_CHECK_ARG(B >= 0, "seq_order: bad B");
    if (B == 0) return ESME_OK;
    ESME_CHECK_ARG(cu_lens && order, "seq_order: null pointer");
    hipLaunchKernelGGL(seq_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cu_lens, B, order);
    return check_launch("seq_order");
}

extern "C" int esme_hip_layernorm(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy,
                                  int64_t T, int E, float eps, void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0, "layernorm: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x && w && y, "layernorm: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= E && ldy >= E, "layernorm: E/ld not multiples of 8");
    ESME_CHECK_ARG(aligned16(x) && aligned16(y) && aligned16(w) && (!b || aligned16(b)), "layernorm: misaligned");
    const dim3 grid((unsigned int)((T + 3) / 4)), block(256);
    const hipStream_t s = (hipStream_t)stream;
#define ESME_LN(N)                                                                                            \
    hipLaunchKernelGGL(layernorm_kernel<N>, grid, block, 0, s, (const u16*)x, ldx, (const u16*)w, (const u16*)b, \
                       (u16*)y, ldy, T, E, eps)
    if (E <= 512) ESME_LN(1);
    else if (E <= 1024) ESME_LN(2);
    else if (E <= 1536) ESME_LN(3);
    else if (E <= 2560) ESME_LN(5);
    else if (E <= 5120) ESME_LN(10);
    else ESME_FAIL(ESME_ERR_UNSUPPORTED, "layernorm: E > 5120 unsupported");
#undef ESME_LN
    return check_launch("layernorm");
}


extern "C" int esme_hip_residual_f32(float* x32, int64_t ld32, const void* o, int64_t ldo, float alpha, int init, void* x16,
                                     int64_t ld16, float* sums, int64_t T, int E, void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0, "residual_f32: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x32 && o && x16, "residual_f32: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && ld32 % 4 == 0 && ldo % 8 == 0 && ld16 % 8 == 0 && ld32 >= E && ldo >= E && ld16 >= E,
                   "residual_f32: E / row strides not multiples of 8");
    ESME_CHECK_ARG(aligned16(x32) && aligned16(o) && aligned16(x16) && (!sums || (reinterpret_cast<uintptr_t>(sums) & 7u) == 0),
                   "residual_f32: misaligned");
    const dim3 grid((unsigned int)((T + 3) / 4)), block(256);
    const hipStream_t s = (hipStream_t)stream;
#define ESME_RF(N) hipLaunchKernelGGL(residual_f32_kernel<N>, grid, block, 0, s, x32, ld32, (const u16*)o, ldo, alpha, init, \
                                      (u16*)x16, ld16, (f32x2*)sums, T, E)
    if (E <= 512) ESME_RF(1);
    else if (E <= 1024) ESME_RF(2);
    else if (E <= 1536) ESME_RF(3);
    else if (E <= 2560) ESME_RF(5);
    else if (E <= 5120) ESME_RF(10);
    else ESME_FAIL(ESME_ERR_UNSUPPORTED, "residual_f32: E > 5120 unsupported");
#undef ESME_RF
    return check_launch("residual_f32");
}

extern "C" int esme_hip_stream_operand_guarded(const float* x32, int64_t ld32, void* x16, int64_t ld16, int64_t lo_off, int f16, const float* scale,
                                               const int32_t* ext_sel, int ext_n, int64_t ext_off, float* sums, uint32_t* col_absmax, int64_t T, int E,
                                               void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0, "stream_operand: bad sizes");
    ESME_CHECK_ARG(!col_absmax || (lo_off != 0 && aligned16(col_absmax)), "stream_operand: col_absmax belongs to the pair form and must be 16-byte aligned");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x32 && x16, "stream_operand: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && ld32 % 4 == 0 && ld16 % 8 == 0 && ld32 >= E && ld16 >= E, "stream_operand: E / row strides not multiples of 8");
    ESME_CHECK_ARG(lo_off == 0 || (lo_off >= E && lo_off % 8 == 0 && ld16 >= lo_off + E), "stream_operand: lo_off must be a multiple of 8 with E <= lo_off <= ld16 - E");
    ESME_CHECK_ARG(aligned16(x32) && aligned16(x16) && (!sums || (reinterpret_cast<uintptr_t>(sums) & 7u) == 0), "stream_operand: misaligned");
    ESME_CHECK_ARG(!scale || (lo_off != 0 && aligned16(scale)), "stream_operand: a column scale belongs to the pair form (lo_off != 0) and must be 16-byte aligned");
    ESME_CHECK_ARG(ext_off == 0 || (lo_off != 0 && ext_off >= E && ext_off + 64 <= lo_off && ext_n >= 0 && ext_n <= 64 && (ext_n == 0 || ext_sel)),
                   "stream_operand: the extension tile is 64 columns between hi and lo (E <= ext_off, ext_off + 64 <= lo_off) with <= 64 selected channels");
    const dim3 grid((unsigned int)((T + 3) / 4)), block(256);
    const hipStream_t s = (hipStream_t)stream;
#define ESME_SO(N) do { if (f16) hipLaunchKernelGGL((stream_operand_kernel<N, true>), grid, block, 0, s, x32, ld32, (u16*)x16, ld16, lo_off, scale, ext_sel, ext_n, ext_off, (f32x2*)sums, col_absmax, T, E); \
                        else hipLaunchKernelGGL((stream_operand_kernel<N, false>), grid, block, 0, s, x32, ld32, (u16*)x16, ld16, lo_off, scale, ext_sel, ext_n, ext_off, (f32x2*)sums, col_absmax, T, E); } while (0)
    if (E <= 512) ESME_SO(1);
    else if (E <= 1024) ESME_SO(2);
    else if (E <= 1536) ESME_SO(3);
    else if (E <= 2560) ESME_SO(5);
    else if (E <= 5120) ESME_SO(10);
    else ESME_FAIL(ESME_ERR_UNSUPPORTED, "stream_operand: E > 5120 unsupported");
#undef ESME_SO
    return check_launch("stream_operand");
}

extern "C" int esme_hip_stream_operand_scaled(const float* x32, int64_t ld32, void* x16, int64_t ld16, int64_t lo_off, int f16, const float* scale,
                                              const int32_t* ext_sel, int ext_n, int64_t ext_off, float* sums, int64_t T, int E, void* stream) {
    return esme_hip_stream_operand_guarded(x32, ld32, x16, ld16, lo_off, f16, scale, ext_sel, ext_n, ext_off, sums, nullptr, T, E, stream);
}

extern "C" int esme_hip_stream_operand(const float* x32, int64_t ld32, void* x16, int64_t ld16, int64_t lo_off, int f16, float* sums,
                                       int64_t T, int E, void* stream) {
    return esme_hip_stream_operand_scaled(x32, ld32, x16, ld16, lo_off, f16, nullptr, nullptr, 0, 0, sums, T, E, stream);
}

extern "C" int esme_hip_pair_to_f32(const void* x, int64_t ld, int64_t lo_off, int f16, float* out, int64_t ld32, int64_t T, int E, void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0, "pair_to_f32: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x && out, "pair_to_f32: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && ld % 8 == 0 && lo_off % 8 == 0 && lo_off >= E && ld >= lo_off + E && ld32 % 4 == 0 && ld32 >= E, "pair_to_f32: bad layout");
    ESME_CHECK_ARG(aligned16(x) && aligned16(out), "pair_to_f32: misaligned");
    const int chunks = E / 8;
    const dim3 grid(grid_for(T * chunks, 256)), block(256);
    if (f16) hipLaunchKernelGGL(pair_to_f32_kernel<true>, grid, block, 0, (hipStream_t)stream, (const u16*)x, ld, lo_off, out, ld32, T, chunks);
    else hipLaunchKernelGGL(pair_to_f32_kernel<false>, grid, block, 0, (hipStream_t)stream, (const u16*)x, ld, lo_off, out, ld32, T, chunks);
    return check_launch("pair_to_f32");
}

extern "C" int esme_hip_layernorm_f32(const float* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy,
                                      int64_t T, int E, float eps, void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0, "layernorm_f32: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x && w && y, "layernorm_f32: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && ldx % 4 == 0 && ldy % 8 == 0 && ldx >= E && ldy >= E, "layernorm_f32: E/ld not multiples of 8");
    ESME_CHECK_ARG(aligned16(x) && aligned16(y) && aligned16(w) && (!b || aligned16(b)), "layernorm_f32: misaligned");
    const dim3 grid((unsigned int)((T + 3) / 4)), block(256);
    const hipStream_t s = (hipStream_t)stream;
#define ESME_LNF(N) hipLaunchKernelGGL(layernorm_f32_kernel<N>, grid, block, 0, s, x, ldx, (const u16*)w, (const u16*)b, (u16*)y, ldy, T, E, eps)
    if (E <= 512) ESME_LNF(1);
    else if (E <= 1024) ESME_LNF(2);
    else if (E <= 1536) ESME_LNF(3);
    else if (E <= 2560) ESME_LNF(5);
    else if (E <= 5120) ESME_LNF(10);
    else ESME_FAIL(ESME_ERR_UNSUPPORTED, "layernorm_f32: E > 5120 unsupported");
#undef ESME_LNF
    return check_launch("layernorm_f32");
}

extern "C" int esme_hip_layernorm_split_checked(const void* x, int64_t ldx, int in_pair, int64_t in_off, const void* w, const void* b, void* y,
                                        int64_t ldy, int64_t out_off, float* y32, int64_t ld32, int64_t T, int E, float eps, int* overflow_flag,
                                        void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0, "layernorm_split: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x && w && y, "layernorm_split: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && ldy % 8 == 0 && out_off % 8 == 0 && out_off >= E && ldy >= out_off + E, "layernorm_split: bad output layout");
    if (in_pair) ESME_CHECK_ARG(ldx % 8 == 0 && in_off % 8 == 0 && in_off >= E && ldx >= in_off + E, "layernorm_split: bad pair input layout");
    else ESME_CHECK_ARG(ldx % 4 == 0 && ldx >= E, "layernorm_split: bad fp32 input layout");
    ESME_CHECK_ARG(aligned16(x) && aligned16(y) && aligned16(w) && (!b || aligned16(b)) && (!y32 || (aligned16(y32) && ld32 % 4 == 0 && ld32 >= E)),
                   "layernorm_split: misaligned");
    const dim3 grid((unsigned int)((T + 3) / 4)), block(256);
    const hipStream_t s = (hipStream_t)stream;
#define ESME_LNS(N) do { if (in_pair == 2) hipLaunchKernelGGL((layernorm_split_kernel<N, 2>), grid, block, 0, s, x, ldx, in_off, (const u16*)w, (const u16*)b, (u16*)y, ldy, out_off, y32, ld32, T, E, eps, overflow_flag); \
                         else if (in_pair) hipLaunchKernelGGL((layernorm_split_kernel<N, 1>), grid, block, 0, s, x, ldx, in_off, (const u16*)w, (const u16*)b, (u16*)y, ldy, out_off, y32, ld32, T, E, eps, overflow_flag); \
                         else hipLaunchKernelGGL((layernorm_split_kernel<N, 0>), grid, block, 0, s, x, ldx, in_off, (const u16*)w, (const u16*)b, (u16*)y, ldy, out_off, y32, ld32, T, E, eps, overflow_flag); } while (0)
    if (E <= 512) ESME_LNS(1);
    else if (E <= 1024) ESME_LNS(2);
    else if (E <= 1536) ESME_LNS(3);
    else if (E <= 2560) ESME_LNS(5);
    else if (E <= 5120) ESME_LNS(10);
    else ESME_FAIL(ESME_ERR_UNSUPPORTED, "layernorm_split: E > 5120 unsupported");
#undef ESME_LNS
    return check_launch("layernorm_split");
}

extern "C" int esme_hip_layernorm_split(const void* x, int64_t ldx, int in_pair, int64_t in_off, const void* w, const void* b, void* y,
                                        int64_t ldy, int64_t out_off, float* y32, int64_t ld32, int64_t T, int E, float eps, void* stream) {
    return esme_hip_layernorm_split_checked(x, ldx, in_pair, in_off, w, b, y, ldy, out_off, y32, ld32, T, E, eps, nullptr, stream);
}

static int rotary_split_impl(const bool f16, void* x, int64_t ld, int64_t lo_off, const float* cosT, const float* sinT, const int32_t* pos,
                                     int64_t T, int nheads, int d, int max_len, void* stream) {
    ESME_CHECK_ARG(T >= 0 && nheads > 0 && d > 0 && max_len > 0, "rotary_split: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x && cosT && sinT && pos, "rotary_split: null pointer");
    if (d % 16 != 0) ESME_FAIL(ESME_ERR_UNSUPPORTED, "rotary_split: head dim must be a multiple of 16");
    ESME_CHECK_ARG(ld % 8 == 0 && lo_off % 8 == 0 && lo_off >= (int64_t)nheads * d && ld >= lo_off + (int64_t)nheads * d, "rotary_split: bad row stride / pair offset");
    ESME_CHECK_ARG(aligned16(x) && aligned16(cosT) && aligned16(sinT), "rotary_split: misaligned");
    const int64_t items = T * nheads * (d / 16);
    if (f16) hipLaunchKernelGGL(rotary_split_kernel<true>, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream, (u16*)x, ld, lo_off, cosT, sinT, pos,
                       T, nheads, d, max_len);
    else hipLaunchKernelGGL(rotary_split_kernel<false>, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream, (u16*)x, ld, lo_off, cosT, sinT, pos,
                       T, nheads, d, max_len);
    return check_launch("rotary_split");
}

extern "C" int esme_hip_rotary_split(void* x, int64_t ld, int64_t lo_off, const float* cosT, const float* sinT, const int32_t* pos,
                                     int64_t T, int nheads, int d, int max_len, void* stream) {
    return rotary_split_impl(false, x, ld, lo_off, cosT, sinT, pos, T, nheads, d, max_len, stream);
}
extern "C" int esme_hip_rotary_split_f16(void* x, int64_t ld, int64_t lo_off, const float* cosT, const float* sinT, const int32_t* pos,
                                         int64_t T, int nheads, int d, int max_len, void* stream) {
    return rotary_split_impl(true, x, ld, lo_off, cosT, sinT, pos, T, nheads, d, max_len, stream);
}

extern "C" int esme_hip_softmax_rows_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t T, int V, int log_flag, void* stream) {
    ESME_CHECK_ARG(T >= 0 && V > 0, "softmax_rows_f32: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x && y && ldx >= V && ldy >= V, "softmax_rows_f32: null pointer or bad stride");
    if (V > 64) ESME_FAIL(ESME_ERR_UNSUPPORTED, "softmax_rows_f32: V > 64 unsupported");
    hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3((unsigned int)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, T, V, log_flag);
    return check_launch("softmax_rows_f32");
}

extern "C" int esme_hip_row_sums(const void* x, int64_t ldx, int64_t T, int E, float* sums, void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0, "row_sums: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x && sums && E % 8 == 0 && ldx % 8 == 0 && ldx >= E && aligned16(x) &&
                   (reinterpret_cast<uintptr_t>(sums) & 7u) == 0, "row_sums: null/misaligned pointer or E, ldx not multiples of 8");
    const dim3 grid((unsigned int)((T + 3) / 4)), block(256);
    const hipStream_t s = (hipStream_t)stream;
#define ESME_RS(N) hipLaunchKernelGGL(row_sums_kernel<N>, grid, block, 0, s, (const u16*)x, ldx, T, E, (f32x2*)sums)
    if (E <= 512) ESME_RS(1);
    else if (E <= 1024) ESME_RS(2);
    else if (E <= 1536) ESME_RS(3);
    else if (E <= 2560) ESME_RS(5);
    else if (E <= 5120) ESME_RS(10);
    else ESME_FAIL(ESME_ERR_UNSUPPORTED, "row_sums: E > 5120 unsupported");
#undef ESME_RS
    return check_launch("row_sums");
}

static int rotary_impl(void* q, void* k, int64_t ld, const void* cosT, const void* sinT,
                       const int32_t* pos, int64_t T, int H, int d, int max_len, bool f16, void* stream) {
    ESME_CHECK_ARG(T >= 0 && H > 0 && d > 0 && max_len > 0, "rotary: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(q && k && cosT && sinT && pos, "rotary: null pointer");
    if (d % 16 != 0) ESME_FAIL(ESME_ERR_UNSUPPORTED, "rotary: head dim must be a multiple of 16");
    ESME_CHECK_ARG(ld % 8 == 0 && ld >= (int64_t)H * d, "rotary: bad row stride");
    ESME_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(cosT) && aligned16(sinT), "rotary: misaligned");
    const int64_t items = T * 2 * H * (d / 16);
    if (f16) hipLaunchKernelGGL(rotary_kernel<true>, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream, (u16*)q,
                                (u16*)k, ld, (const u16*)cosT, (const u16*)sinT, pos, T, H, d, max_len);
    else hipLaunchKernelGGL(rotary_kernel<false>, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream, (u16*)q,
                            (u16*)k, ld, (const u16*)cosT, (const u16*)sinT, pos, T, H, d, max_len);
    return check_launch("rotary");
}

extern "C" int esme_hip_rotary_varlen(void* q, void* k, int64_t ld, const void* cosT, const void* sinT,
                                      const int32_t* pos, int64_t T, int H, int d, int max_len, void* stream) {
    return rotary_impl(q, k, ld, cosT, sinT, pos, T, H, d, max_len, false, stream);
}

extern "C" int esme_hip_rotary_varlen_f16(void* q, void* k, int64_t ld, const void* cosT, const void* sinT,
                                          const int32_t* pos, int64_t T, int H, int d, int max_len, void* stream) {
    return rotary_impl(q, k, ld, cosT, sinT, pos, T, H, d, max_len, true, stream);
}

extern "C" int esme_hip_softmax_rows(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t T, int V,
                                     int log_flag, void* stream) {
    ESME_CHECK_ARG(T >= 0 && V > 0, "softmax_rows: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(x && y && ldx >= V && ldy >= V, "softmax_rows: null pointer or bad stride");
    if (V > 64) ESME_FAIL(ESME_ERR_UNSUPPORTED, "softmax_rows: V > 64 unsupported");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned int)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)x, ldx, (u16*)y, ldy, T, V, log_flag);
    return check_launch("softmax_rows");
}

static int gather_scatter(const void* src, const int64_t* idx, void* dst, int64_t n, int E, int64_t bound, void* stream,
                          int scatter) {
    ESME_CHECK_ARG(n >= 0 && E > 0 && bound >= 0, "gather/scatter_rows: bad sizes");
    if (n == 0) return ESME_OK;
    ESME_CHECK_ARG(src && idx && dst, "gather/scatter_rows: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && aligned16(src) && aligned16(dst), "gather/scatter_rows: E %% 8 != 0 or misaligned");
    const int chunks = E / 8;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * chunks, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)src, idx, (u32x4*)dst, n, chunks, scatter, bound);
    return check_launch(scatter ? "scatter_rows" : "gather_rows");
}

extern "C" int esme_hip_gather_rows(const void* src, int64_t src_rows, const int64_t* idx, void* dst, int64_t n, int E,
                                    void* stream) {
    return gather_scatter(src, idx, dst, n, E, src_rows, stream, 0);
}
extern "C" int esme_hip_scatter_rows(const void* src, const int64_t* idx, void* dst, int64_t dst_rows, int64_t n, int E,
                                     void* stream) {
    return gather_scatter(src, idx, dst, n, E, dst_rows, stream, 1);
}

extern "C" int esme_hip_segment_mean(const void* x, int64_t ldx, const int32_t* cu_lens, int B, int E, void* out,
                                     int64_t ldo, int dtype_f32, void* stream) {
    ESME_CHECK_ARG(B >= 0 && E > 0, "segment_mean: bad sizes");
    if (B == 0) return ESME_OK;
    ESME_CHECK_ARG(x && cu_lens && out && ldx >= E && ldo >= E, "segment_mean: null pointer or bad stride");
    const int vec = dtype_f32 ? 4 : 8;
    ESME_CHECK_ARG(E % 8 == 0 && ldx % vec == 0 && ldo % vec == 0 && aligned16(x) && aligned16(out),
                   "segment_mean: E %% 8 != 0 or misaligned rows");
    const dim3 grid((unsigned int)B, (unsigned int)((E + 63) / 64));
    if (dtype_f32)
        hipLaunchKernelGGL(segment_mean_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, cu_lens, E, out, ldo);
    else
        hipLaunchKernelGGL(segment_mean_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, cu_lens, E, out, ldo);
    return check_launch("segment_mean");
}

static int qk_norm_rotary_impl(void* q, void* k, int64_t ld, const void* wq, const void* wk, const void* bq,
                               const void* bk, float eps, const void* cosT, const void* sinT,
                               const int32_t* pos, int64_t T, int heads, int head_dim, int max_len, float q_scale,
                               bool f16, void* stream, uint32_t* qk_sumsq = nullptr) {
    ESME_CHECK_ARG(T >= 0 && heads > 0 && head_dim > 0 && max_len > 0, "qk_norm_rotary: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(q && k && wq && wk && cosT && sinT && pos, "qk_norm_rotary: null pointer");
    if (head_dim != 16 && head_dim != 32 && head_dim != 64 && head_dim != 128)
        ESME_FAIL(ESME_ERR_UNSUPPORTED, "qk_norm_rotary: head dim must be 16, 32, 64 or 128");
    const int64_t E64 = (int64_t)heads * head_dim;
    if (E64 > 5120) ESME_FAIL(ESME_ERR_UNSUPPORTED, "qk_norm_rotary: heads * head_dim > 5120 unsupported");
    const int E = (int)E64;
    ESME_CHECK_ARG(ld % 8 == 0 && ld >= E, "qk_norm_rotary: bad row stride");
    ESME_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(wq) && aligned16(wk) && (!bq || aligned16(bq)) &&
                   (!bk || aligned16(bk)) && aligned16(cosT) && aligned16(sinT), "qk_norm_rotary: misaligned");
#ifndef ESME_QKN_RPW
#define ESME_QKN_RPW 2
#endif
    constexpr int RPW = ESME_QKN_RPW;                        // rows per wave
    const dim3 grid((unsigned int)((T + 2 * RPW - 1) / (2 * RPW))), block(256);
    const hipStream_t s = (hipStream_t)stream;
#define ESME_QKN(N)                                                                                                 \
    do { if (f16) hipLaunchKernelGGL((qk_norm_rotary_kernel<N, RPW, true>), grid, block, 0, s, (u16*)q, (u16*)k, ld, (const u16*)wq, (const u16*)wk, \
                       (const u16*)bq, (const u16*)bk, eps, (const u16*)cosT, (const u16*)sinT, pos, T, E, head_dim, max_len, q_scale, qk_sumsq); \
    else hipLaunchKernelGGL((qk_norm_rotary_kernel<N, RPW>), grid, block, 0, s, (u16*)q, (u16*)k, ld, (const u16*)wq, (const u16*)wk, \
                       (const u16*)bq, (const u16*)bk, eps, (const u16*)cosT, (const u16*)sinT, pos, T, E, head_dim, max_len, q_scale, (unsigned int*)nullptr); } while (0)
    if (E <= 512) ESME_QKN(1);
    else if (E <= 1024) ESME_QKN(2);
    else if (E <= 1536) ESME_QKN(3);
    else if (E <= 2560) ESME_QKN(5);
    else ESME_QKN(10);
#undef ESME_QKN
    return check_launch("qk_norm_rotary");
}

extern "C" int esme_hip_qk_norm_rotary_scaled(void* q, void* k, int64_t ld, const void* wq, const void* wk, const void* bq,
                                              const void* bk, float eps, const void* cosT, const void* sinT,
                                              const int32_t* pos, int64_t T, int heads, int head_dim, int max_len, float q_scale,
                                              void* stream) {
    return qk_norm_rotary_impl(q, k, ld, wq, wk, bq, bk, eps, cosT, sinT, pos, T, heads, head_dim, max_len, q_scale, false, stream);
}

extern "C" int esme_hip_qk_norm_rotary_f16(void* q, void* k, int64_t ld, const void* wq, const void* wk, const void* bq,
                                           const void* bk, float eps, const void* cosT, const void* sinT,
                                           const int32_t* pos, int64_t T, int heads, int head_dim, int max_len, void* stream) {
    return qk_norm_rotary_impl(q, k, ld, wq, wk, bq, bk, eps, cosT, sinT, pos, T, heads, head_dim, max_len, 1.0f, true, stream);
}

extern "C" int esme_hip_qk_norm_rotary_f16_guarded(void* q, void* k, int64_t ld, const void* wq, const void* wk, const void* bq,
                                                   const void* bk, float eps, const void* cosT, const void* sinT,
                                                   const int32_t* pos, int64_t T, int heads, int head_dim, int max_len, uint32_t* qk_sumsq, void* stream) {
    ESME_CHECK_ARG(!qk_sumsq || (reinterpret_cast<uintptr_t>(qk_sumsq) & 3u) == 0, "qk_norm_rotary: misaligned qk_sumsq");
    return qk_norm_rotary_impl(q, k, ld, wq, wk, bq, bk, eps, cosT, sinT, pos, T, heads, head_dim, max_len, 1.0f, true, stream, qk_sumsq);
}

extern "C" int esme_hip_qk_norm_rotary_f16_scaled(void* q, void* k, int64_t ld, const void* wq, const void* wk, const void* bq,
                                                  const void* bk, float eps, const void* cosT, const void* sinT,
                                                  const int32_t* pos, int64_t T, int heads, int head_dim, int max_len, float q_scale, uint32_t* qk_sumsq, void* stream) {
    ESME_CHECK_ARG(!qk_sumsq || (reinterpret_cast<uintptr_t>(qk_sumsq) & 3u) == 0, "qk_norm_rotary: misaligned qk_sumsq");
    ESME_CHECK_ARG(q_scale > 0.f && q_scale == q_scale, "qk_norm_rotary: q_scale must be positive");
    return qk_norm_rotary_impl(q, k, ld, wq, wk, bq, bk, eps, cosT, sinT, pos, T, heads, head_dim, max_len, q_scale, true, stream, qk_sumsq);
}

extern "C" int esme_hip_qk_norm_rotary(void* q, void* k, int64_t ld, const void* wq, const void* wk, const void* bq,
                                       const void* bk, float eps, const void* cosT, const void* sinT,
                                       const int32_t* pos, int64_t T, int heads, int head_dim, int max_len, void* stream) {
    return esme_hip_qk_norm_rotary_scaled(q, k, ld, wq, wk, bq, bk, eps, cosT, sinT, pos, T, heads, head_dim, max_len, 1.0f, stream);
}

extern "C" int esme_hip_embed_positions(const int64_t* tokens, const void* table, const void* pos_table,
                                        const int32_t* pos_idx, int pos_offset, void* out, int64_t T, int E, int V,
                                        int P, int mask_idx, void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0 && V > 0 && P > 0, "embed_positions: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(tokens && table && pos_table && pos_idx && out, "embed_positions: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && aligned16(table) && aligned16(pos_table) && aligned16(out),
                   "embed_positions: E %% 8 != 0 or misaligned");
    const int chunks = E / 8;
    hipLaunchKernelGGL(embed_pos_kernel, dim3(grid_for(T * chunks, 256)), dim3(256), 0, (hipStream_t)stream, tokens,
                       (const u32x4*)table, (const u32x4*)pos_table, pos_idx, pos_offset, (u32x4*)out, (float*)nullptr, T, chunks, V, P,
                       mask_idx);
    return check_launch("embed_positions");
}

extern "C" int esme_hip_embed_positions_f32(const int64_t* tokens, const void* table, const void* pos_table, const int32_t* pos_idx,
                                            int pos_offset, float* out, int64_t T, int E, int V, int P, int mask_idx, void* stream) {
    ESME_CHECK_ARG(T >= 0 && E > 0 && V > 0 && P > 0, "embed_positions_f32: bad sizes");
    if (T == 0) return ESME_OK;
    ESME_CHECK_ARG(tokens && table && pos_table && pos_idx && out, "embed_positions_f32: null pointer");
    ESME_CHECK_ARG(E % 8 == 0 && aligned16(table) && aligned16(pos_table) && aligned16(out), "embed_positions_f32: E %% 8 != 0 or misaligned");
    const int chunks = E / 8;
    hipLaunchKernelGGL(embed_pos_kernel, dim3(grid_for(T * chunks, 256)), dim3(256), 0, (hipStream_t)stream, tokens,
                       (const u32x4*)table, (const u32x4*)pos_table, pos_idx, pos_offset, (u32x4*)nullptr, out, T, chunks, V, P, mask_idx);
    return check_launch("embed_positions_f32");
}
