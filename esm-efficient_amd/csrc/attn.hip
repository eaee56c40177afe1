// Varlen (cu_seqlens-indexed, block-diagonal) multi-head self-attention forward for
// gfx950:  per sequence i and head h,  O = softmax(Q K^T * scale) V, non-causal.
//
// Workgroup = 4 waves = one 128-row query tile of one (sequence, head); each wave owns 32
// query rows and walks the sequence's keys in tiles of 64.
//
// Both contractions run on v_mfma_f32_32x32x16_bf16 in TRANSPOSED form so that the
// softmax axis (keys) lies along a lane's registers and the query index is the lane:
//     S^T (key x q) = K (key x d) . Q^T (d x q)         A = K rows from LDS, B = Q rows (registers)
//     O^T (d x q)   = V^T (d x key) . P^T (key x q)      A = V^T rows from LDS, B = P (registers)
// Row max / row sum are then in-lane reductions plus ONE exchange with lane^32, the online
// softmax rescale factor is lane-local for both S^T and O^T, and P never leaves registers:
// the K rows fed to the first MFMA are permuted (bits 2<->3 of the row index) so that the
// 8 scores a lane holds per 16-key step are 8 CONSECUTIVE keys, i.e. exactly the B-operand
// layout of the second MFMA, with V^T read as one ds_read_b128 per fragment.
//
// LDS: K tile [64 keys][D] (16-B chunks XOR-swizzled against the row index, conflict-free
// for the 32-row fragment reads) and V^T tile [D][64 keys] (each thread transposes a 4x4
// bf16 block in registers while staging).  Global loads for tile t+1 are issued before the
// MFMAs of tile t (register-staged, written to LDS after the barrier).
#include "common.h"
#include "launch.h"

namespace esme {

static constexpr int QT = 128;   // query rows per workgroup (4 waves x 32)
static constexpr int KT = 64;    // keys per tile

struct AttnArgs {
    const u16* q; const u16* k; const u16* v; int64_t ld;
    u16* o; int64_t ldo;
    const int32_t* cu;
    int H;
    float scale_log2;            // softmax_scale * log2(e)
};

// swizzle of the 16-byte chunk index inside a K-tile row of D bf16 (CPR chunks per row)
template <int D>
__device__ __forceinline__ int kswz(int row) {
    constexpr int CPR = D / 8;                   // 2, 4, 8, 16
    constexpr int RPB = 16 / CPR;                // rows per 256-B bank row: 8, 4, 2, 1
    return (row / RPB) & (CPR - 1);
}

template <int D>
__global__ __launch_bounds__(256) void attn_varlen_kernel(const AttnArgs a) {
    constexpr int DS = D / 16;                   // k-steps of the QK^T contraction
    constexpr int DB = (D + 31) / 32;            // 32-row blocks of O^T
    constexpr int CPR = D / 8;                   // 16-B chunks per K row
    constexpr int KCH = KT * CPR;                // chunks in a K tile
    constexpr int KI = (KCH + 255) / 256;        // K chunks per thread
    constexpr int VB = (KT / 4) * (D / 4);       // 4x4 blocks in a V tile
    constexpr int VI = (VB + 255) / 256;
    constexpr int K_BYTES = KT * D * 2;

    __shared__ __attribute__((aligned(16))) char smem[K_BYTES + D * 128];
    char* const Ks = smem;
    char* const Vt = smem + K_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int s0 = a.cu[b], S = a.cu[b + 1] - s0;
    const int q0 = blockIdx.x * QT;
    if (q0 >= S) return;

    const int64_t ld = a.ld;
    const u16* qb = a.q + (int64_t)s0 * ld + h * D;
    const u16* kb = a.k + (int64_t)s0 * ld + h * D;
    const u16* vb = a.v + (int64_t)s0 * ld + h * D;

    // ---- Q fragments (B operand of S^T): lane (q = l31, hi) holds Q[q][ds*16 + hi*8 .. +7]
    const int qrow = q0 + wave * 32 + l31;
    const bool wave_active = (q0 + wave * 32) < S;       // wave-uniform
    const int qrow_c = qrow < S ? qrow : S - 1;
    bf16x8 qf[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds)
        qf[ds] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)qrow_c * ld + ds * 16 + hi * 8);

    // ---- staging assignments
    u32x4 kreg[KI];
    u32x2 vreg[VI][4];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int c = i * 256 + tid;
            if (KCH >= 256 || c < KCH) {
                int row = kv0 + c / CPR;
                row = row < S ? row : S - 1;
                kreg[i] = *reinterpret_cast<const u32x4*>(kb + (int64_t)row * ld + (c % CPR) * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int blk = i * 256 + tid;
            if (VB >= 256 || blk < VB) {
                const int dq = blk % (D / 4), kq = blk / (D / 4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    int row = kv0 + kq * 4 + kk;
                    row = row < S ? row : S - 1;
                    vreg[i][kk] = *reinterpret_cast<const u32x2*>(vb + (int64_t)row * ld + dq * 4);
                }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int c = i * 256 + tid;
            if (KCH >= 256 || c < KCH) {
                const int row = c / CPR, ch = c % CPR;
                *reinterpret_cast<u32x4*>(Ks + row * (D * 2) + ((ch ^ kswz<D>(row)) << 4)) = kreg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int blk = i * 256 + tid;
            if (VB >= 256 || blk < VB) {
                const int dq = blk % (D / 4), kq = blk / (D / 4);
                // 4x4 transpose of 16-bit elements: in[kk] = {d0d1, d2d3} of key kk
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const int w = dd >> 1;
                    unsigned int e0, e1, e2, e3;
                    if (dd & 1) {
                        e0 = vreg[i][0][w] >> 16; e1 = vreg[i][1][w] & 0xffff0000u;
                        e2 = vreg[i][2][w] >> 16; e3 = vreg[i][3][w] & 0xffff0000u;
                    } else {
                        e0 = vreg[i][0][w] & 0xffffu; e1 = vreg[i][1][w] << 16;
                        e2 = vreg[i][2][w] & 0xffffu; e3 = vreg[i][3][w] << 16;
                    }
                    const int drow = dq * 4 + dd;
                    const int ch = kq >> 1;                         // 16-B chunk (8 keys) of the V^T row
                    u32x2 out = {e0 | e1, e2 | e3};
                    *reinterpret_cast<u32x2*>(Vt + drow * 128 + ((ch ^ ((drow >> 1) & 7)) << 4) + (kq & 1) * 8) = out;
                }
            }
        }
    };

    // K row fed to MFMA row i of key block kbk: bits 2 and 3 of i swapped
    const int krow_perm = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | (l31 & 16);

    f32x16 oacc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c = a.scale_log2;

    const int ntiles = (S + KT - 1) / KT;
    load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * KT;
        __syncthreads();                      // all waves done reading the previous tile
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(kv0 + KT);
        if (!wave_active) continue;

        // ---- S^T = K . Q^T for two 32-key blocks
        f32x16 sacc[2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kbk][r] = 0.f;
            const int row = kbk * 32 + krow_perm;
            const char* rp = Ks + row * (D * 2);
            const int sw = kswz<D>(row);
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(rp + (((ds * 2 + hi) ^ sw) << 4));
                sacc[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], sacc[kbk], 0, 0, 0);
            }
        }
        // register r of block kbk holds key kv0 + kbk*32 + 16*(r>>3) + 8*hi + (r&7)
        const bool tail = kv0 + KT > S;
        float tmax = -1e30f;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (tail) {
                    const int key = kv0 + kbk * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= S) sacc[kbk][r] = -1e30f;
                }
                tmax = fmaxf(tmax, sacc[kbk][r]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    p[j] = exp2f(fmaf(sacc[kbk][8 * s + j], c, -mc));
                    psum += p[j];
                }
                u32x4 pk = pack8(p);
                pf[kbk][s] = __builtin_bit_cast(bf16x8, pk);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int i = 0; i < DB; ++i) {
            int drow = i * 32 + l31;
            if (D < 32) drow &= (D - 1);          // D = 16: upper lanes re-read valid rows, results discarded
            const char* rp = Vt + drow * 128;
            const int sw = (drow >> 1) & 7;
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(rp + (((kbk * 4 + s * 2 + hi) ^ sw) << 4));
                    oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kbk][s], oacc[i], 0, 0, 0);
                }
        }
    }

    if (!wave_active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < S) {
        u16* op = a.o + (int64_t)(s0 + qrow) * a.ldo + h * D;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = i * 32 + 8 * g + 4 * hi;
                if (d < D) {
                    u32x2 pk = {pack_bf16(oacc[i][4 * g] * inv, oacc[i][4 * g + 1] * inv),
                                pack_bf16(oacc[i][4 * g + 2] * inv, oacc[i][4 * g + 3] * inv)};
                    *reinterpret_cast<u32x2*>(op + d) = pk;
                }
            }
    }
}

}  // namespace esme

using namespace esme;

extern "C" int esme_hip_attn_varlen_fwd(const void* q, const void* k, const void* v, int64_t ld_qkv, void* o,
                                        int64_t ld_o, const int32_t* cu_lens, int B, int64_t T, int H, int d,
                                        int max_len, float softmax_scale, void* stream) {
    ESME_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && d > 0 && max_len >= 0, "attn: bad sizes");
    if (T == 0 || B == 0) return ESME_OK;
    ESME_CHECK_ARG(q && k && v && o && cu_lens, "attn: null pointer");
    ESME_CHECK_ARG(ld_qkv % 8 == 0 && ld_qkv >= (int64_t)H * d && ld_o % 4 == 0 && ld_o >= (int64_t)H * d,
                   "attn: bad row strides");
    ESME_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && (reinterpret_cast<uintptr_t>(o) & 7u) == 0,
                   "attn: misaligned");
    ESME_CHECK_ARG(max_len > 0 && H <= 65535 && B <= 65535, "attn: max_len must be > 0, H and B <= 65535");
    AttnArgs a{(const u16*)q, (const u16*)k, (const u16*)v, ld_qkv, (u16*)o, ld_o, cu_lens, H,
               softmax_scale * 1.4426950408889634f};
    const dim3 grid((unsigned int)((max_len + QT - 1) / QT), (unsigned int)H, (unsigned int)B), block(256);
    const hipStream_t s = (hipStream_t)stream;
    switch (d) {
        case 16: hipLaunchKernelGGL(attn_varlen_kernel<16>, grid, block, 0, s, a); break;
        case 32: hipLaunchKernelGGL(attn_varlen_kernel<32>, grid, block, 0, s, a); break;
        case 64: hipLaunchKernelGGL(attn_varlen_kernel<64>, grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL(attn_varlen_kernel<128>, grid, block, 0, s, a); break;
        default: ESME_FAIL(ESME_ERR_UNSUPPORTED, "attn: head dim must be 16, 32, 64 or 128");
    }
    return check_launch("attn_varlen_fwd");
}
