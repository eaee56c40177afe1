// Varlen (cu_seqlens-indexed, block-diagonal) multi-head self-attention forward for
// gfx950:  per sequence i and head h,  O = softmax(Q K^T * scale) V, non-causal.
//
// Workgroup = 4 waves = one query tile of 128 * QB rows of one (sequence, head); each wave owns
// QB blocks of 32 query rows (QB = 2 when sequences are long enough: every K / V^T fragment read
// from LDS then feeds two MFMAs, the two blocks' MFMA chains and softmax VALU work interleave
// inside the wave, and staging + barrier cost per query row halves) and walks the sequence's keys
// in tiles of 64.
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
// LDS (double-buffered, one barrier per tile): K tile [64 keys][D] (16-B chunks XOR-swizzled
// against the row index, conflict-free for the 32-row fragment reads) and V^T tile [D][64 keys]
// (each thread transposes a 4x4 bf16 block in registers while staging; the lane->block map
// makes the 8-byte transposed writes conflict-free too).  Global loads for tile t+2 are issued
// while tile t+1 is written to LDS and tile t is in the MFMAs.  Online softmax uses exp2 with
// the scale folded in and a defer-max threshold (rescale O only when a row max grew by more
// than 2^8), so the common tile does no O-wide VALU pass.
#include "common.h"
#include "launch.h"

namespace esme {

static constexpr int QT = 128;   // query rows per workgroup per q-block (4 waves x 32)
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

template <int D, int QB>
__global__ __launch_bounds__(256, 2) void attn_varlen_kernel(const AttnArgs a) {
    constexpr int DS = D / 16;                   // k-steps of the QK^T contraction
    constexpr int DB = (D + 31) / 32;            // 32-row blocks of O^T
    constexpr int CPR = D / 8;                   // 16-B chunks per K row
    constexpr int KCH = KT * CPR;                // chunks in a K tile
    constexpr int KI = (KCH + 255) / 256;        // K chunks per thread
    constexpr int DQ = D / 4;                    // 4-wide column groups of V
    constexpr int DQ_HI_BITS = (D == 16 ? 0 : D == 32 ? 1 : D == 64 ? 2 : 3);
    constexpr int VI = (16 * DQ + 255) / 256;    // 4x4 V blocks per thread
    constexpr int K_BYTES = KT * D * 2;
    constexpr int BUF = K_BYTES + D * 128;       // one K tile + one V^T tile
    constexpr float THR = 8.0f;                  // defer-max threshold (log2 units)

    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int s0 = a.cu[b], S = a.cu[b + 1] - s0;
    const int q0 = blockIdx.x * (QT * QB);
    if (q0 >= S) return;

    const unsigned int ld = (unsigned int)a.ld;
    const u16* qb = a.q + (int64_t)s0 * a.ld + h * D;
    const u16* kb = a.k + (int64_t)s0 * a.ld + h * D;
    const u16* vb = a.v + (int64_t)s0 * a.ld + h * D;

    // ---- Q fragments (B operand of S^T): lane (q = l31, hi) holds Q[q][ds*16 + hi*8 .. +7];
    // wave w owns rows q0 + w*32*QB + b*32 + l31 for its q-blocks b = 0..QB-1
    int qrow[QB];
    bool blk_active[QB];                                   // wave-uniform
    bf16x8 qf[QB][DS];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int r0 = q0 + (wave * QB + b) * 32;
        qrow[b] = r0 + l31;
        blk_active[b] = r0 < S;
        const unsigned int qc = qrow[b] < S ? qrow[b] : S - 1;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds)
            qf[b][ds] = *reinterpret_cast<const bf16x8*>(qb + (qc * ld + ds * 16 + hi * 8));
    }
    const bool wave_active = blk_active[0];

    // ---- staging assignments.  K: chunk c -> (row c / CPR, chunk c % CPR).
    // V: each thread owns 4x4 (key x d) blocks; lane bits are laid out so that the 16
    // lanes of a ds_write_b64 group cover 4 column groups x 4 key groups = 16 distinct
    // 8-byte positions of the swizzled 128-B V^T rows (conflict-free), while a wave's
    // global load still reads whole 128-B lines.
    unsigned int koff[KI];
    int krow[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
        const int c = i * 256 + tid;
        krow[i] = c / CPR;
        koff[i] = (unsigned int)(c % CPR) * 8u;
    }
    int v_dq[VI], v_kq[VI];
#pragma unroll
    for (int i = 0; i < VI; ++i) {
        const int rest = (lane >> 4) | (wave << 2) | (i << 4);
        v_dq[i] = (lane & 3) | ((rest & ((1 << DQ_HI_BITS) - 1)) << 2);
        v_kq[i] = ((lane >> 2) & 3) | ((rest >> DQ_HI_BITS) << 2);
    }
    u32x4 kreg[KI];
    u32x2 vreg[VI][4];
    auto load_tile = [&](int kv0) {
        const bool full = kv0 + KT <= S;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            if (KCH >= 256 * KI || krow[i] < KT) {
                unsigned int row = kv0 + krow[i];
                if (!full) row = row < (unsigned int)S ? row : S - 1;
                kreg[i] = *reinterpret_cast<const u32x4*>(kb + (row * ld + koff[i]));
            }
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            if (v_kq[i] < 16) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    unsigned int row = kv0 + v_kq[i] * 4 + kk;
                    if (!full) row = row < (unsigned int)S ? row : S - 1;
                    vreg[i][kk] = *reinterpret_cast<const u32x2*>(vb + (row * ld + v_dq[i] * 4));
                }
            }
        }
    };
    auto store_tile = [&](char* Ks, char* Vt) {
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            if (KCH >= 256 * KI || krow[i] < KT) {
                const int row = krow[i], ch = (int)(koff[i] >> 3);
                *reinterpret_cast<u32x4*>(Ks + row * (D * 2) + ((ch ^ kswz<D>(row)) << 4)) = kreg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            if (v_kq[i] < 16) {
                const int dq = v_dq[i], kq = v_kq[i];
                // 4x4 transpose of 16-bit elements: in[kk] = {d0d1, d2d3} of key kk
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const int w = dd >> 1;
                    unsigned int e0, e1;
                    if (dd & 1) {
                        e0 = __builtin_amdgcn_perm(vreg[i][1][w], vreg[i][0][w], 0x07060302u);
                        e1 = __builtin_amdgcn_perm(vreg[i][3][w], vreg[i][2][w], 0x07060302u);
                    } else {
                        e0 = __builtin_amdgcn_perm(vreg[i][1][w], vreg[i][0][w], 0x05040100u);
                        e1 = __builtin_amdgcn_perm(vreg[i][3][w], vreg[i][2][w], 0x05040100u);
                    }
                    const int drow = dq * 4 + dd;
                    const int ch = kq >> 1;                         // 16-B chunk (8 keys) of the V^T row
                    u32x2 out = {e0, e1};
                    *reinterpret_cast<u32x2*>(Vt + drow * 128 + ((ch ^ ((drow >> 1) & 7)) << 4) + (kq & 1) * 8) = out;
                }
            }
        }
    };

    // K row fed to MFMA row i of a key block: bits 2 and 3 of i swapped
    const int krow_perm = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | (l31 & 16);

    f32x16 oacc[QB][DB];
    float mc[QB], l_run[QB];           // running max (already multiplied by c: log2 units) and row sum
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        mc[b] = -1e30f; l_run[b] = 0.f;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[b][i][r] = 0.f;
    }
    const float c = a.scale_log2;

    const int ntiles = (S + KT - 1) / KT;
    load_tile(0);
    store_tile(smem, smem + K_BYTES);
    if (ntiles > 1) load_tile(KT);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * KT;
        const char* Ks = smem + (t & 1) * BUF;
        const char* Vt = Ks + K_BYTES;
        if (wave_active) {
            // ---- S^T = K . Q^T for two 32-key blocks; each K fragment feeds all QB q-blocks
            f32x16 sacc[QB][2];
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
                for (int b = 0; b < QB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[b][kbk][r] = 0.f;
                const int row = kbk * 32 + krow_perm;
                const char* rp = Ks + row * (D * 2);
                const int sw = kswz<D>(row);
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(rp + (((ds * 2 + hi) ^ sw) << 4));
#pragma unroll
                    for (int b = 0; b < QB; ++b)
                        sacc[b][kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[b][ds], sacc[b][kbk], 0, 0, 0);
                }
            }
            bf16x8 pf[QB][2][2];
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                // register r of block kbk holds key kv0 + kbk*32 + 16*(r>>3) + 8*hi + (r&7)
                if (kv0 + KT > S) {
#pragma unroll
                    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kv0 + kbk * 32 + 16 * (r >> 3) + 8 * hi + (r & 7) >= S) sacc[b][kbk][r] = -1e30f;
                }
                float tmax = sacc[b][0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sacc[b][0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[b][1][r]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float tmc = tmax * c;
                // defer-max: rescale only when some row's max grew by more than THR (in log2 units);
                // otherwise keep the old reference max -- P is then bounded by 2^THR, which fp32 sums
                // and the bf16 P (same relative precision at any scale) absorb.  Wave-uniform branch.
                if (__any(tmc > mc[b] + THR)) {
                    const float mn = fmaxf(mc[b], tmc);
                    const float alpha = __builtin_amdgcn_exp2f(mc[b] - mn);
                    mc[b] = mn;
                    l_run[b] *= alpha;
#pragma unroll
                    for (int i = 0; i < DB; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[b][i][r] *= alpha;
                }
                float psum = 0.f;
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float p[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            p[j] = __builtin_amdgcn_exp2f(fmaf(sacc[b][kbk][8 * s + j], c, -mc[b]));
                            psum += p[j];
                        }
                        u32x4 pk = pack8(p);
                        pf[b][kbk][s] = __builtin_bit_cast(bf16x8, pk);
                    }
                l_run[b] += psum;
            }

            // ---- O^T += V^T . P^T; each V^T fragment feeds all QB q-blocks
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
#pragma unroll
                        for (int b = 0; b < QB; ++b)
                            oacc[b][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[b][kbk][s], oacc[b][i], 0, 0, 0);
                    }
            }
        }
        // stage tile t+1 (loaded into registers while tile t was computed) into the other buffer;
        // that buffer was last read for tile t-1, which every wave finished before the barrier
        // that ended iteration t-1.
        if (t + 1 < ntiles) {
            char* Kn = smem + ((t + 1) & 1) * BUF;
            store_tile(Kn, Kn + K_BYTES);
            if (t + 2 < ntiles) load_tile(kv0 + 2 * KT);
        }
        __syncthreads();
    }

    if (!wave_active) return;
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32, 64);
        const float inv = 1.0f / l_tot;
        if (qrow[b] < S) {
            u16* op = a.o + (int64_t)(s0 + qrow[b]) * a.ldo + h * D;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = i * 32 + 8 * g + 4 * hi;
                    if (d < D) {
                        u32x2 pk = {pack_bf16(oacc[b][i][4 * g] * inv, oacc[b][i][4 * g + 1] * inv),
                                    pack_bf16(oacc[b][i][4 * g + 2] * inv, oacc[b][i][4 * g + 3] * inv)};
                        *reinterpret_cast<u32x2*>(op + d) = pk;
                    }
                }
        }
    }
}

}  // namespace esme

using namespace esme;

static int g_force_qb = 0;      // test hook: force q-blocks per wave (0 = heuristic)
extern "C" void esme_hip_debug_set_attn_qb(int v) { g_force_qb = v; }

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
    // two 32-row q-blocks per wave when the longest sequence fills at least one 256-row tile
    // (head dim 128 keeps one: its accumulators alone take 128 VGPRs per q-block)
    const int qb = (g_force_qb ? g_force_qb : (max_len >= 192 ? 2 : 1));
    const bool two = qb == 2 && d <= 64;
    const int rows = QT * (two ? 2 : 1);
    const dim3 grid((unsigned int)((max_len + rows - 1) / rows), (unsigned int)H, (unsigned int)B), block(256);
    const hipStream_t s = (hipStream_t)stream;
#define ESME_ATTN(DD)                                                                         \
    case DD:                                                                                   \
        if (two) hipLaunchKernelGGL((attn_varlen_kernel<DD, (DD <= 64 ? 2 : 1)>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((attn_varlen_kernel<DD, 1>), grid, block, 0, s, a);            \
        break;
    switch (d) {
        ESME_ATTN(16)
        ESME_ATTN(32)
        ESME_ATTN(64)
        ESME_ATTN(128)
        default: ESME_FAIL(ESME_ERR_UNSUPPORTED, "attn: head dim must be 16, 32, 64 or 128");
    }
#undef ESME_ATTN
    return check_launch("attn_varlen_fwd");
}
