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
#ifndef ESME_ATTN_CM4_ASM
#define ESME_ATTN_CM4_ASM 0
#endif
#include "gemm.h"

#ifndef ESME_ATTN_ABL           // lab builds only (timing ablations of attn_pp64_kernel, WRONG results; profiles/r05_attn_rowsum_ablation.txt):
#define ESME_ATTN_ABL 0         // 1 = the two row-sum v_add per score pair dropped; 3 = ... and 4 MFMAs per phase on a ones fragment issued instead
#endif                          // (what "row sums on the matrix pipe" would execute: VERDICT r4 item 3b)
#ifndef ESME_ATTN_DMA0          // MFMA slots of a phase behind which this wave's two LDS-DMA pieces are issued (A/B builds)
#define ESME_ATTN_DMA0 3
#define ESME_ATTN_DMA1 11
#endif
#include <atomic>
#include <type_traits>

namespace esme {

static constexpr int QT = 128;   // query rows per workgroup per q-block (4 waves x 32)
static constexpr int KT = 64;    // keys per tile

struct AttnArgs {
    const u16* q; const u16* k; const u16* v; int64_t ld;
    u16* o; int64_t ldo;
    const int32_t* cu;
    int H;
    float scale_log2;            // softmax_scale * log2(e)
    int nqt;                     // ping-pong kernel: query tiles per sequence
    int nhb;                     // ping-pong kernel: H * B (sequence, head) pairs
    float thr;                   // defer-max threshold in log2 units (0 = rescale whenever a row max grows)
    int spec;                    // ping-pong kernel: speculative softmax after a row's first key tile (see the kernel header)
    const int32_t* order;        // optional: work item i belongs to sequence order[i] (longest first: esme_hip_seq_order); NULL = i
};

// swizzle of the 16-byte chunk index inside a K-tile row of D bf16 (CPR chunks per row)
template <int D>
__device__ __forceinline__ int kswz(int row) {
    constexpr int CPR = D / 8;                   // 2, 4, 8, 16
    constexpr int RPB = 16 / CPR;                // rows per 256-B bank row: 8, 4, 2, 1
    return (row / RPB) & (CPR - 1);
}

// F16 (precision 'half'): q, k, v, P and o are IEEE fp16; the host then runs the classic online softmax with exact maxima (P <= 1).
template <int D, int QB, bool F16 = false>
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
    const float THR = a.thr;                     // defer-max threshold (log2 units)

    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = a.order ? a.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y;
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
                ESME_LDS_CHECK(Ks + row * (D * 2) + ((ch ^ kswz<D>(row)) << 4), 16, smem, 2 * BUF);
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
                    ESME_LDS_CHECK(Vt + drow * 128 + ((ch ^ ((drow >> 1) & 7)) << 4) + (kq & 1) * 8, 8, smem, 2 * BUF);
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
                    ESME_LDS_CHECK(rp + (((ds * 2 + hi) ^ sw) << 4), 16, smem, 2 * BUF);
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(rp + (((ds * 2 + hi) ^ sw) << 4));
#pragma unroll
                    for (int b = 0; b < QB; ++b)
                        sacc[b][kbk] = mfma_32x32x16<F16>(kf, qf[b][ds], sacc[b][kbk]);
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
                        u32x4 pk = pack8t<F16>(p);
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
                        ESME_LDS_CHECK(rp + (((kbk * 4 + s * 2 + hi) ^ sw) << 4), 16, smem, 2 * BUF);
                        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(rp + (((kbk * 4 + s * 2 + hi) ^ sw) << 4));
#pragma unroll
                        for (int b = 0; b < QB; ++b)
                            oacc[b][i] = mfma_32x32x16<F16>(vf, pf[b][kbk][s], oacc[b][i]);
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
                        u32x2 pk = {pack16<F16>(oacc[b][i][4 * g] * inv, oacc[b][i][4 * g + 1] * inv),
                                    pack16<F16>(oacc[b][i][4 * g + 2] * inv, oacc[b][i][4 * g + 3] * inv)};
                        *reinterpret_cast<u32x2*>(op + d) = pk;
                    }
                }
        }
    }
}


// =============================================================================================
// Split-operand ('exact') mode: the same contraction with every MFMA operand carried as a (hi, lo) bf16 pair
// (x = hi + lo to ~2^-17 relative):
//     S  = Q K^T  ~  Qh Kh^T + Qh Kl^T + Ql Kh^T                       (3 MFMA passes; the Ql Kl^T term is 2^-18 relative)
//     O  = P V    ~  Ph Vh   + Ph Vl   + Pl Vh        with  Ph = bf16(P), Pl = bf16(P - Ph)  formed in registers
// fp32 scores, classic online softmax with every row maximum exact, fp32 row sums of the UNROUNDED P, and the result leaves as
// a pair as well (the out-projection's K-doubled operand).  q / k / v: hi at the given pointer, lo `lo_in` elements further
// right in the same row (the QKV projection's pair epilogue); o: lo at `lo_out`.  Structure of attn_varlen_kernel<D, 1> (one
// 32-row query block per wave, register-staged double-buffered tiles, V transposed while staging) with both halves of the
// K / V^T tiles side by side in LDS.  An accuracy mode: 3x the MFMA work of the fast kernels, not tuned beyond that.
struct AttnSplitArgs {
    AttnArgs a;
    int64_t lo_in, lo_out;
};

// QKP (precision 'half' on a model whose attention scores are large, round 5): IEEE fp16 operands with ONLY q and k as pairs --
//     S = Qh Kh^T + Qh Kl^T + Ql Kh^T   (scores to ~2^-21 of |q||k|: an fp16 q / k alone costs 2^-12 |q||k|, i.e. several tenths of a
//     score unit once the projections of massive stream channels push |score| into the hundreds),   O = P V  single pass, fp16 P, V, O.
template <int D, bool F16 = false, bool QKP = false>
__global__ __launch_bounds__(256, 2) void attn_split_kernel(const AttnSplitArgs sa) {
    static_assert(!QKP || F16, "q/k-only pairs: the fp16 form");
    constexpr int NPT = QKP ? 1 : 2;                   // V (and P, O) parts
    const AttnArgs& a = sa.a;
    constexpr int DS = D / 16;
    constexpr int DB = (D + 31) / 32;
    constexpr int CPR = D / 8;
    constexpr int KCH = KT * CPR;
    constexpr int KI = (KCH + 255) / 256;
    constexpr int DQ = D / 4;
    constexpr int DQ_HI_BITS = (D == 16 ? 0 : D == 32 ? 1 : D == 64 ? 2 : 3);
    constexpr int VI = (16 * DQ + 255) / 256;
    constexpr int K_BYTES = KT * D * 2;
    constexpr int V_BYTES = D * 128;
    constexpr int BUF = 2 * K_BYTES + NPT * V_BYTES;   // [K hi | K lo | V^T hi | V^T lo]  (QKP: one V^T)
    static_assert(D == 16 || D == 32 || D == 64 || D == 128, "split-operand attention: head dims 16, 32, 64, 128");

    extern __shared__ __attribute__((aligned(16))) char smem_split[];
    char* smem = smem_split;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = a.order ? a.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y;
    const int s0 = a.cu[b], S = a.cu[b + 1] - s0;
    const int q0 = blockIdx.x * QT;
    if (q0 >= S) return;

    const unsigned int ld = (unsigned int)a.ld;
    const u16* qb = a.q + (int64_t)s0 * a.ld + h * D;
    const u16* kb = a.k + (int64_t)s0 * a.ld + h * D;
    const u16* vb = a.v + (int64_t)s0 * a.ld + h * D;

    const int qrow = q0 + wave * 32 + l31;
    const bool wave_active = q0 + wave * 32 < S;
    bf16x8 qf[2][DS];
    {
        const unsigned int qc = qrow < S ? qrow : S - 1;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int ds = 0; ds < DS; ++ds)
                qf[pt][ds] = *reinterpret_cast<const bf16x8*>(qb + pt * sa.lo_in + (qc * ld + ds * 16 + hi * 8));
    }

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
    u32x4 kreg[2][KI];
    u32x2 vreg[NPT][VI][4];
    auto load_tile = [&](int kv0) {
        const bool full = kv0 + KT <= S;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                if (KCH >= 256 * KI || krow[i] < KT) {
                    unsigned int row = kv0 + krow[i];
                    if (!full) row = row < (unsigned int)S ? row : S - 1;
                    kreg[pt][i] = *reinterpret_cast<const u32x4*>(kb + pt * sa.lo_in + (row * ld + koff[i]));
                }
            }
            if (pt >= NPT) continue;
#pragma unroll
            for (int i = 0; i < VI; ++i) {
                if (v_kq[i] < 16) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        unsigned int row = kv0 + v_kq[i] * 4 + kk;
                        if (!full) row = row < (unsigned int)S ? row : S - 1;
                        vreg[pt][i][kk] = *reinterpret_cast<const u32x2*>(vb + pt * sa.lo_in + (row * ld + v_dq[i] * 4));
                    }
                }
            }
        }
    };
    auto store_tile = [&](char* buf) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            char* Ks = buf + pt * K_BYTES;
            char* Vt = buf + 2 * K_BYTES + pt * V_BYTES;
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                if (KCH >= 256 * KI || krow[i] < KT) {
                    const int row = krow[i], ch = (int)(koff[i] >> 3);
                    ESME_LDS_CHECK(Ks + row * (D * 2) + ((ch ^ kswz<D>(row)) << 4), 16, smem, 2 * BUF);
                    *reinterpret_cast<u32x4*>(Ks + row * (D * 2) + ((ch ^ kswz<D>(row)) << 4)) = kreg[pt][i];
                }
            }
            if (pt >= NPT) continue;
#pragma unroll
            for (int i = 0; i < VI; ++i) {
                if (v_kq[i] < 16) {
                    const int dq = v_dq[i], kq = v_kq[i];
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        const int w = dd >> 1;
                        unsigned int e0, e1;
                        if (dd & 1) {
                            e0 = __builtin_amdgcn_perm(vreg[pt][i][1][w], vreg[pt][i][0][w], 0x07060302u);
                            e1 = __builtin_amdgcn_perm(vreg[pt][i][3][w], vreg[pt][i][2][w], 0x07060302u);
                        } else {
                            e0 = __builtin_amdgcn_perm(vreg[pt][i][1][w], vreg[pt][i][0][w], 0x05040100u);
                            e1 = __builtin_amdgcn_perm(vreg[pt][i][3][w], vreg[pt][i][2][w], 0x05040100u);
                        }
                        const int drow = dq * 4 + dd;
                        const int ch = kq >> 1;
                        u32x2 out = {e0, e1};
                        ESME_LDS_CHECK(Vt + drow * 128 + ((ch ^ ((drow >> 1) & 7)) << 4) + (kq & 1) * 8, 8, smem, 2 * BUF);
                        *reinterpret_cast<u32x2*>(Vt + drow * 128 + ((ch ^ ((drow >> 1) & 7)) << 4) + (kq & 1) * 8) = out;
                    }
                }
            }
        }
    };

    const int krow_perm = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | (l31 & 16);

    f32x16 oacc[DB];
    float mc = -1e30f, l_run = 0.f;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    const float c = a.scale_log2;

    const int ntiles = (S + KT - 1) / KT;
    load_tile(0);
    store_tile(smem);
    if (ntiles > 1) load_tile(KT);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * KT;
        const char* buf = smem + (t & 1) * BUF;
        if (wave_active) {
            f32x16 sacc[2];
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kbk][r] = 0.f;
                const int row = kbk * 32 + krow_perm;
                const char* rp = buf + row * (D * 2);
                const int sw = kswz<D>(row);
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) {
                    const bf16x8 kh = *reinterpret_cast<const bf16x8*>(rp + (((ds * 2 + hi) ^ sw) << 4));
                    ESME_LDS_CHECK(rp + K_BYTES + (((ds * 2 + hi) ^ sw) << 4), 16, smem, 2 * BUF);
                    const bf16x8 kl = *reinterpret_cast<const bf16x8*>(rp + K_BYTES + (((ds * 2 + hi) ^ sw) << 4));
                    sacc[kbk] = mfma_32x32x16<F16>(kl, qf[0][ds], sacc[kbk]);      // small terms first
                    sacc[kbk] = mfma_32x32x16<F16>(kh, qf[1][ds], sacc[kbk]);
                    sacc[kbk] = mfma_32x32x16<F16>(kh, qf[0][ds], sacc[kbk]);
                }
            }
            if (kv0 + KT > S) {
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kv0 + kbk * 32 + 16 * (r >> 3) + 8 * hi + (r & 7) >= S) sacc[kbk][r] = -1e30f;
            }
            float tmax = sacc[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sacc[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[1][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float tmc = tmax * c;
            if (__any(tmc > mc)) {                         // every row maximum exact
                const float mn = fmaxf(mc, tmc);
                const float alpha = __builtin_amdgcn_exp2f(mc - mn);
                mc = mn;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            }
            bf16x8 pf[NPT][2][2];                          // [hi / lo][key block][k-step]
            float psum = 0.f;
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float p[8], ph[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        p[j] = __builtin_amdgcn_exp2f(fmaf(sacc[kbk][8 * s + j], c, -mc));
                        psum += p[j];
                    }
                    const u32x4 pk = pack8t<F16>(p);
                    pf[0][kbk][s] = __builtin_bit_cast(bf16x8, pk);
                    if constexpr (!QKP) {
                        unpack8t<F16>(pk, ph);
#pragma unroll
                        for (int j = 0; j < 8; ++j) p[j] -= ph[j];
                        pf[NPT - 1][kbk][s] = __builtin_bit_cast(bf16x8, pack8t<F16>(p));
                    }
                }
            l_run += psum;

            const char* Vt = buf + 2 * K_BYTES;
#pragma unroll
            for (int i = 0; i < DB; ++i) {
                int drow = i * 32 + l31;
                if (D < 32) drow &= (D - 1);
                const char* rp = Vt + drow * 128;
                const int sw = (drow >> 1) & 7;
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 vh = *reinterpret_cast<const bf16x8*>(rp + (((kbk * 4 + s * 2 + hi) ^ sw) << 4));
                        if constexpr (!QKP) {
                            ESME_LDS_CHECK(rp + V_BYTES + (((kbk * 4 + s * 2 + hi) ^ sw) << 4), 16, smem, 2 * BUF);
                            const bf16x8 vl = *reinterpret_cast<const bf16x8*>(rp + V_BYTES + (((kbk * 4 + s * 2 + hi) ^ sw) << 4));
                            oacc[i] = mfma_32x32x16<F16>(vl, pf[0][kbk][s], oacc[i]);
                            oacc[i] = mfma_32x32x16<F16>(vh, pf[NPT - 1][kbk][s], oacc[i]);
                        }
                        oacc[i] = mfma_32x32x16<F16>(vh, pf[0][kbk][s], oacc[i]);
                    }
            }
        }
        if (t + 1 < ntiles) {
            store_tile(smem + ((t + 1) & 1) * BUF);
            if (t + 2 < ntiles) load_tile(kv0 + 2 * KT);
        }
        __syncthreads();
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
                    float o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = oacc[i][4 * g + e] * inv;
                    const u32x2 pk = {pack16<F16>(o4[0], o4[1]), pack16<F16>(o4[2], o4[3])};
                    *reinterpret_cast<u32x2*>(op + d) = pk;
                    if constexpr (!QKP) {
                        const u32x2 pl = {pack16<F16>(o4[0] - lo16<F16>(pk[0]), o4[1] - hi16<F16>(pk[0])), pack16<F16>(o4[2] - lo16<F16>(pk[1]), o4[3] - hi16<F16>(pk[1]))};
                        *reinterpret_cast<u32x2*>(op + sa.lo_out + d) = pl;
                    }
                }
            }
    }
}


// =============================================================================================
// Head dim 64: software-pipelined ("ping-pong") kernel.
//
// Same math and MFMA operand layouts as attn_varlen_kernel above; what changes is the schedule, the data
// path into LDS and the amount of VALU work per score (the first-generation kernel is VALU-bound: ~210
// vector instructions per 32 x 64 block of scores against 16 MFMAs).
//
// * A wave owns TWO 32-row query blocks b0, b1 and runs them half a key tile out of phase, so that every
//   softmax has the 16 MFMAs of the OTHER block issued between its instructions (one MFMA + its LDS
//   fragment reads per 7 VALU instructions, pinned with sched_barrier + empty asm statements):
//
//     phase A(t):  softmax(b0, tile t)   ||   S^T(b1, t)   = K_t     Q_b1^T       (8 MFMA)
//                                             O^T(b1)     += V_t-1^T P(b1, t-1)^T  (8 MFMA)
//     phase B(t):  softmax(b1, tile t)   ||   S^T(b0, t+1) = K_t+1   Q_b0^T       (8 MFMA)
//                                             O^T(b0)     += V_t^T   P(b0, t)^T    (8 MFMA)
//
// * Speculative softmax (spec = 1): after a row's first key tile fixed its reference maximum m, later
//   tiles compute P = exp2(s*c - m) WITHOUT looking for the tile maximum (16 v_max3 + exchange + compare
//   per block) -- fp32 / bf16 hold P up to 2^127, the bf16 rounding of P is relative, O and l accumulate
//   in fp32, so any finite P is as accurate as a P <= 1.  Only overflow must be caught: a row sum that is
//   not < 1e30 (inf / NaN included) makes the WORKGROUP redo its work item with the classic online softmax
//   (same code, need_max on every tile).  spec = 0 is the classic online softmax with the defer-max
//   threshold `thr` (thr = 0: a row's maximum is always exact).
// * K AND V tiles go HBM -> LDS by LDS-DMA (buffer_load ... lds: no VGPR round trip, no ds_write, no
//   transposition pass): K as [key][d] rows for ds_read_b128 fragments, V as [key][d] rows as well -- the
//   V^T fragments of the second MFMA are gathered by ds_read_b64_tr_b16 (the hardware 4x4 transposing
//   read).  The descriptors end at the sequence's last row: rows past the end read as zeros.
// * The DMA instructions are inline asm, NOT the builtin: hipcc guards every LDS read that follows an
//   LDS-DMA it knows of with s_waitcnt vmcnt(0) (possible alias), which made every key tile wait a memory
//   latency for a prefetch that is needed two tiles later.  The kernel counts instead: each wave waits
//   for its own DMAs of the PREVIOUS iteration (s_waitcnt vmcnt(n issued in this one)) right before the
//   one s_barrier that ends a key tile.  The pieces (1 KB per wave instruction, 60-180 cycles of issue
//   each) are issued one at a time from inside the phases, under a running MFMA.
// * What bounds it (tools/lab/mfma_issue_probe.hip, tools/attn_power_probe.py): with 2 waves per SIMD a
//   stream of {1 MFMA, 1-2 LDS reads, 7 VALU of this mix} runs at 48 cycles per MFMA (v_exp_f32 = 8,
//   other VALU = 4 cycles of the SIMD's one VALU port; v_pk_*_f32 are slower than two scalar ops), the
//   loop measures 60; and the kernel runs AT the 1 400 W package power cap (2.0 GHz instead of 2.4; all-zero
//   inputs: 2.39 GHz, +20-25 % throughput), so removed stall cycles come back only in part.
//
// LDS: a ring of 4 slots (K tile + V tile, 16 KB each); iteration t reads K of slots t, t+1 and V of slots
// t-1, t, and prefetches K tile t+3 / V tile t+2; ONE barrier per key tile.  A workgroup is NW waves =
// NW*64 query rows of one (sequence, head): NW = 4 runs two workgroups per CU (one's prologue / epilogue
// overlaps the other's main loop), NW = 8 stays behind the tuning hook.  The row sum exchange with lane^32
// is a v_permlane32_swap (VALU), not an LDS permute; results leave through a wave-private LDS slab as
// whole 128-byte rows (16 B per lane).
// QP ("q prescaled"): q arrives multiplied by softmax_scale * log2(e) (the QKV projection's epilogue does it in fp32 before its
// one bf16 rounding: esme_gemm_fusion_t.q_scale), so a score IS its exponent: the speculative pass computes P = exp2(s) straight
// from the accumulators -- no reference maximum at all, not even on the first tile, and 5 instead of 7 VALU instructions per
// score pair in a loop that is bound by the VALU port (-5 % at S = 500, -8 % at S = 1 002).  fp32 / bf16 hold P from 2^-126 to
// 2^127; a row sum that overflows (>= 1e30) or vanishes (<= 1e-30) flags the work item, which is redone with the classic online
// softmax (the row maximum subtracted before the pipelined region), exactly as for the speculative pass of the plain form.
// Head dim 32 (round 4: ESM2-150M, BASELINE config 2; ESM2-35M through its padded layout): the same kernel with D = 32 -- K / V rows of
// 64 B (tiles of 4 KB, one LDS-DMA piece per wave and tile), two k-steps per S^T block and ONE 32-row block of O^T, i.e. 8 MFMAs per
// phase against the same 16 score pairs: two pairs per MFMA slot.  The VALU port bounds it harder than head dim 64 (the softmax work
// per score is the same, the MFMA work half), but the ping-pong schedule, the LDS-DMA data path and the speculative softmax carry over.
// F16 (precision 'half'): q, k, v, P and o are IEEE fp16 (11 significant bits).  fp16 ends at 65 504, so there is no QP form (P = exp2(score) with
// no reference at all); the speculative pass against the FIRST tile's maximum stays (P = 2^(how far a later score beats that maximum): a
// handful on real data) with the overflow test tightened to fp16's range -- a work item that trips it is redone with exact maxima (P <= 1).
// QP && F16 (round 6): the no-reference form with a FIXED reference of 4 (log2 units) -- the score accumulators start at -4.0 (the C operand of each score block's
// first MFMA: sixteen registers kept for the purpose; as an inline constant of the instruction only behind ESME_ATTN_CM4_ASM, see common.h), so P = 2^(s - 4) stays inside fp16 for scores up to 20 (13.9 in natural units: e^13.9 = 10^6 times the weight of a zero
// score).  A row whose scores go higher trips the overflow test (partial sum >= 3e4), a row whose sum falls below S * 2^-14 (its P values average below fp16's smallest normal
// number) trips the vanished-sum test: both redo the work item with exact maxima, exactly as the bf16 form does at 1e30 / 1e-30.
template <int NW, bool QP = false, int D = 64, bool F16 = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_pp64_kernel(const AttnArgs a) {
    constexpr float S0 = (QP && F16) ? -4.0f : 0.0f;            // where the score accumulators start
    constexpr int DS = D / 16, DB = D / 32, NT = NW * 64;
    constexpr int ROWB = D * 2;                  // bytes per K / V row
    constexpr int K_BYTES = KT * D * 2;          // 8 KB (head dim 64) / 4 KB (32)
    constexpr int SLOT = K_BYTES + D * 128;      // K tile [64 keys][D] + V tile [64 keys][D]
    constexpr int ROWS = NW * 64;
    constexpr int NPV = 4 * DB, NQK = 2 * DS, NM = NPV + NQK;      // MFMAs of a phase: O^T += V^T P^T, then S^T = K Q^T (16 / 8)
    constexpr int PPS = 16 / NM;                 // score pairs per MFMA slot (1 / 2)
    static_assert(D == 64 || D == 32, "head dims 64 and 32");
    static_assert(NW == 4 || (NW == 8 && D == 64), "4 waves (8: head dim 64 only)");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // block id -> work item.  Block i runs on XCD i % 8 (observed dispatch policy; speed only): the nqt query tiles
    // of one (sequence, head) get ids 8 apart -- neighbours in ONE XCD's queue, so the second tile finds K / V in
    // that XCD's L2 -- while consecutive (sequence, head) pairs go round-robin over the XCDs, which balances ragged
    // batches (a contiguous id range per XCD put a whole 2 300-residue protein on one XCD: 1.5x slower).
    const unsigned int xcd = blockIdx.x & 7u, bi = blockIdx.x >> 3;
    const int qt = (int)(bi % (unsigned int)a.nqt);
    const unsigned int hb = (bi / (unsigned int)a.nqt) * 8u + xcd;
    if (hb >= (unsigned int)a.nhb) return;
    const int h = (int)(hb % (unsigned int)a.H), bi_seq = (int)(hb / (unsigned int)a.H);
    const int b = a.order ? a.order[bi_seq] : bi_seq;       // (speed only: the longest sequences' work items are dispatched first)
    const int s0 = a.cu[b], S = a.cu[b + 1] - s0;
    const int q0 = qt * ROWS;
    if (q0 >= S) return;

    const unsigned int ld = (unsigned int)a.ld;
    const u16* qb = a.q + (int64_t)s0 * a.ld + h * D;
    // K / V of this (sequence, head) behind buffer descriptors that end with the head's slice of row S-1
    const unsigned int kv_bytes = ((unsigned int)(S - 1) * ld + D) * 2u;
    // Raw descriptor words (base, stride 0, bytes, DATA_FORMAT 32): the LDS-DMA is issued from inline asm (below).
    auto make_rsrc = [&](const u16* p) -> u32x4 {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        return u32x4{(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v),
                     (unsigned int)__builtin_amdgcn_readfirstlane((int)((unsigned int)(v >> 32) & 0xffffu)),
                     (unsigned int)__builtin_amdgcn_readfirstlane((int)kv_bytes), 0x00020000u};
    };
    const u32x4 krs = make_rsrc(a.k + (int64_t)s0 * a.ld + h * D);
    const u32x4 vrs = make_rsrc(a.v + (int64_t)s0 * a.ld + h * D);
    // One LDS-DMA instruction: 64 lanes x 16 B from (descriptor, per-lane byte offset) to LDS bytes [dst, dst + 1024).
    // Inline asm ON PURPOSE: hipcc guards every LDS read that follows an LDS-DMA it knows about with `s_waitcnt vmcnt(0)`
    // (it cannot prove the read does not alias the DMA's target), i.e. each key tile would wait a full memory latency for
    // the prefetch of a tile that is needed two iterations later.  The kernel does its own accounting instead: counted
    // vmcnt + s_barrier at the end of every key tile.  M0 (the LDS destination) is saved and restored: hipcc owns it.
    auto dma16 = [&](const u32x4 rs, const unsigned int voff, const char* dst) {
        ESME_LDS_CHECK(dst, 1024, smem, 4 * SLOT);
        const unsigned int d = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(uintptr_t)dst);
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(d), "s"(rs) : "memory");
    };

    // ---- Q fragments of the wave's two q-blocks (B operand of S^T): lane (q = l31, hi) holds Q[q][ds*16 + hi*8 ..]
    bf16x8 qf[2][DS];
    const int wrow0 = q0 + wave * 64;                      // first query row of this wave
    const bool wave_active = wrow0 < S;                    // wave-uniform
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
        const int qr = wrow0 + bb * 32 + l31;
        const unsigned int qc = qr < S ? qr : S - 1;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds)
            qf[bb][ds] = *reinterpret_cast<const bf16x8*>(qb + (qc * ld + ds * 16 + hi * 8));
    }

    // ---- staging.  K and V tiles both go HBM -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, 1 KB = 8 rows per
    // wave instruction, no VGPR round trip, no ds_write): the LDS image is lane-linear, so the chunk swizzle is applied
    // to the per-lane SOURCE address; rows past the sequence end read as zeros (descriptor bounds).  V stays row-major
    // ([key][d], as in HBM): the V^T fragments of the second MFMA are gathered by ds_read_b64_tr_b16 (below).
    constexpr int KI = K_BYTES / (NW * 1024);                    // DMA instructions per wave per K (or V) tile (2 or 1)
    constexpr int CPR = D / 8;                                   // 16-B chunks per row; a 1 KB piece = 64 / CPR rows
    const unsigned int tile_bytes = (unsigned int)KT * ld * 2u;
    unsigned int kg0, vg0;                                       // byte offset of this lane's chunk inside a tile (piece 0)
    {
        const int r = wave * (64 / CPR) + lane / CPR, pch = lane % CPR;    // LDS row / chunk position this lane fills
        kg0 = ((unsigned int)r * ld + ((pch ^ kswz<D>(r)) * 8)) * 2u;
        // V, head dim 64: 64-B halves swapped on rows 2, 3 (mod 4); head dim 32: rows of 64 B as they are (the 32 lanes of a transposing
        // read cover 4 rows x 64 B = one whole 256-B bank row)
        vg0 = ((unsigned int)r * ld + ((D == 64 ? (pch ^ (((r >> 1) & 1) << 2)) : pch) * 8)) * 2u;
    }
    const unsigned int kg_step = (unsigned int)(NW * (64 / CPR)) * ld * 2u;   // piece i: rows + NW * 64 / CPR (same swizzles: a multiple of 16 rows)
    auto dma_k = [&](int tile, char* slot) {
        const unsigned int base = (unsigned int)tile * tile_bytes + kg0;
#pragma unroll
        for (int i = 0; i < KI; ++i)
            dma16(krs, base + i * kg_step, slot + (i * NW + wave) * 1024);
    };
    auto dma_v = [&](int tile, char* slot) {
        const unsigned int base = (unsigned int)tile * tile_bytes + vg0;
#pragma unroll
        for (int i = 0; i < KI; ++i)
            dma16(vrs, base + i * kg_step, slot + K_BYTES + (i * NW + wave) * 1024);
    };

    // ---- per-lane LDS fragment offsets.  K row fed to MFMA row i: bits 2 and 3 of i swapped (P lands in the
    // B-operand layout of the second MFMA); the chunk swizzles do not depend on the 32-row block.
    const int krow_perm = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | (l31 & 16);
    int kfo[DS];
#pragma unroll
    for (int i = 0; i < DS; ++i) kfo[i] = krow_perm * ROWB + (((i * 2 + hi) ^ kswz<D>(krow_perm)) << 4);
    // V^T fragment (A operand of O^T += V^T P^T: row = d, k = key) of 32-d block db, 16-key step ks: two transposing
    // reads of 4 keys each.  ds_read_b64_tr_b16 works on 16-lane groups: lane 4j + p of a group SUPPLIES the 8 bytes
    // V[key j][4p .. 4p+3] of a [4 keys][16 d] block, lane c RECEIVES column c (V[key 0..3][c]).  Lane (l31, hi) of the
    // MFMA wants d = l31, keys hi*8 + 0..7: group (lane >> 4) & 1 covers d 0..15 / 16..31, so this lane supplies row
    // hi*8 + j (+4 for the second read), bytes gsel*32 + p*8 of the 64-B half that holds block db (halves swapped on rows
    // with bit 1 set: the four 64-B row pieces of a 32-lane group fall into four different 16-bank quarters).
    int vb[DB];
    {
        const int j = (lane & 15) >> 2, p = lane & 3, gsel = (lane >> 4) & 1;
#pragma unroll
        for (int db = 0; db < DB; ++db) vb[db] = K_BYTES + (hi * 8 + j) * ROWB + (D == 64 ? ((db ^ (j >> 1)) * 64) : 0) + gsel * 32 + p * 8;
    }
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    auto vfrag = [&](const char* Vs, const int db, const int ks) -> bf16x8 {
        typedef __attribute__((address_space(3))) s16x4* ltr_t;
        const char* p = Vs + vb[db] + ks * (16 * ROWB);
        ESME_LDS_CHECK(p, 8, smem, 4 * SLOT); ESME_LDS_CHECK(p + 4 * ROWB, 8, smem, 4 * SLOT);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(p));
        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(p + 4 * ROWB));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 oacc[2][DB], sacc[2][2];
#if ESME_ATTN_ABL & 2
    f32x16 abl_l = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // ONE set for both blocks (a real kernel needs two: +32 registers)
#endif
    u32x4 pw[2][2][2];                 // P of block bb as packed bf16: [bb][32-key block][16-key step]
    float mc[2], lrun[2];
    const float c = a.scale_log2, thr = a.thr;
    int ovf = 0;                       // a speculative softmax of this wave overflowed

    // the 16 MFMAs a phase issues for block bm: m < 8: O^T(bm) += V^T P^T; m >= 8: S^T(bm) against the K tile at Ks.
    // O^T first: P(bm) dies half-way through the phase, and the new scores of bm are first written when the first half of
    // block bs's old scores has been consumed (their registers can be reused; the loop is register-bound).
    // Consecutive MFMAs alternate between the two accumulators of a contraction (32-row blocks of O^T, 32-key blocks of
    // S^T): an MFMA never follows one on the same accumulator with VALU instructions in between (a 43-cycle cliff).
    auto frag = [&](int m, const char* Ks, const char* Vs) -> bf16x8 {
        if (m < NPV) return vfrag(Vs, m % DB, m / DB);
        ESME_LDS_CHECK(Ks + ((m - NPV) & 1) * (32 * ROWB) + kfo[(m - NPV) >> 1], 16, smem, 4 * SLOT);
        return *reinterpret_cast<const bf16x8*>(Ks + ((m - NPV) & 1) * (32 * ROWB) + kfo[(m - NPV) >> 1]);
    };

    // QP && F16: sixteen registers of -4.0, made opaque so that they are kept (hipcc would otherwise rebuild the splat with 16 v_mov in front of every use)
    f32x16 zneg = {S0, S0, S0, S0, S0, S0, S0, S0, S0, S0, S0, S0, S0, S0, S0, S0};
    if constexpr (QP && F16) asm volatile("" : "+v"(zneg));
    // One phase: softmax of block BS on its finished scores, interleaved with the 16 MFMAs of block BM.
    // SPEC: speculative softmax (no tile maximum; see the header).  TAIL: mask the keys past the sequence end.
    auto phase = [&](auto BS_, auto BM_, const bool need_max, const bool tail, const char* Ks, const char* Vs, const int kv0, auto&& hook) __attribute__((always_inline)) {
        constexpr int bs = decltype(BS_)::value, bm = decltype(BM_)::value;
        bf16x8 fr[3];
        fr[0] = frag(0, Ks, Vs);
        fr[1] = frag(1, Ks, Vs);
        auto mfma_step = [&](int m) {
            if (m + 2 < NM) fr[(m + 2) % 3] = frag(m + 2, Ks, Vs);
            if (m >= NPV) {
                const int j = m - NPV, kbk = j & 1, ds = j >> 1;
                if (ds == 0) {
                    if constexpr (QP && F16) {
#if ESME_ATTN_CM4_ASM
                        sacc[bm][kbk] = mfma_32x32x16_f16_cm4(fr[m % 3], qf[bm][ds]);     // starts at S0 = -4 (inline constant)
#else
                        sacc[bm][kbk] = mfma_32x32x16<F16>(fr[m % 3], qf[bm][ds], zneg);   // starts at S0 = -4 (a register block kept for the purpose)
#endif
                    } else {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        sacc[bm][kbk] = mfma_32x32x16<F16>(fr[m % 3], qf[bm][ds], z);
                    }
                } else {
                    sacc[bm][kbk] = mfma_32x32x16<F16>(fr[m % 3], qf[bm][ds], sacc[bm][kbk]);
                }
            } else {
                const int db = m % DB, ks = m / DB;
                oacc[bm][db] = mfma_32x32x16<F16>(
                    fr[m % 3], __builtin_bit_cast(bf16x8, pw[bm][ks >> 1][ks & 1]), oacc[bm][db]);
#if ESME_ATTN_ABL & 2
                if (db == DB - 1) {          // the row-sum MFMA of this 16-key step: ones (32 x 16) times P^T
                    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                    abl_l = mfma_32x32x16<F16>(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, pw[bm][ks >> 1][ks & 1]), abl_l);
                }
#endif
            }
        };
        // P, packed to bf16, and four partial row sums of block bs against the reference maximum -nm.  Pair p covers
        // scores (kbk, r), (kbk, r + 1).  The three steps of a pair -- x = s*c - m (2 v_fma), P = exp2(x) (2 v_exp),
        // sum + pack (2 v_add + 1 v_cvt_pk) -- are SKEWED over three consecutive MFMA slots (slot m: fma of pair m+1, exp
        // of pair m, sum/pack of pair m-1), so no instruction of a slot waits for another of the same slot: an in-order
        // wave otherwise stalls on the fma -> exp -> add chain (transcendental latency) in every slot.  The empty asm pins
        // the slot's instructions where they are written (between two MFMAs): hipcc otherwise sinks the softmax below
        // the last MFMA of the phase, next to its consumers.
        float ps0, ps1, ps2, ps3;
        float xa0, xa1, pa0 = 0.f, pa1 = 0.f;               // in flight: x of the next pair, P of the previous pair
        auto pair_fma = [&](const int p, const float nm) {
            const int kbk = p >> 3, r = (2 * p) & 15;
            xa0 = fmaf(sacc[bs][kbk][r], c, nm);
            xa1 = fmaf(sacc[bs][kbk][r + 1], c, nm);
        };
        auto pair_sum_pack = [&](const int p, const float q0, const float q1) {
            const int kbk = p >> 3, r = (2 * p) & 15;
            pw[bs][kbk][r >> 3][(r & 7) >> 1] = pack16<F16>(q0, q1);
#if !(ESME_ATTN_ABL & 1)
            if (p & 1) { ps2 += q0; ps3 += q1; } else { ps0 += q0; ps1 += q1; }
#endif
        };
        auto softmax_slot = [&](const int m, const float nm) {      // pair step m = 0..15 (one per MFMA slot at head dim 64, two at 32)
            const float q0 = pa0, q1 = pa1;                  // P of pair m - 1
            if constexpr (QP) {                              // the scores are the exponents (the exact pass has subtracted the maximum already)
                const int kb2 = m >> 3, r2 = (2 * m) & 15;
                pa0 = __builtin_amdgcn_exp2f(sacc[bs][kb2][r2]);
                pa1 = __builtin_amdgcn_exp2f(sacc[bs][kb2][r2 + 1]);
                if (m >= 1) pair_sum_pack(m - 1, q0, q1);
                const int pk = m >= 1 ? m - 1 : 0, kbk = pk >> 3, r = (2 * pk) & 15;
                asm volatile("" : "+v"(pw[bs][kbk][r >> 3]), "+v"(ps0), "+v"(ps1), "+v"(ps2), "+v"(ps3), "+v"(pa0), "+v"(pa1));
            } else {
                const float x0 = xa0, x1 = xa1;              // x of pair m (from slot m - 1)
                if (m + 1 < 16) pair_fma(m + 1, nm);
                pa0 = __builtin_amdgcn_exp2f(x0);
                pa1 = __builtin_amdgcn_exp2f(x1);
                if (m >= 1) pair_sum_pack(m - 1, q0, q1);
                const int pk = m >= 1 ? m - 1 : 0, kbk = pk >> 3, r = (2 * pk) & 15;
                asm volatile("" : "+v"(pw[bs][kbk][r >> 3]), "+v"(ps0), "+v"(ps1), "+v"(ps2), "+v"(ps3), "+v"(xa0), "+v"(xa1), "+v"(pa0), "+v"(pa1));
            }
        };
        // exact tile maximum of block bs's rows (both key halves), in log2 units
        auto tile_max = [&]() -> float {
            float tmax = sacc[bs][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sacc[bs][0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[bs][1][r]);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * c;
        };
        auto rescale_to = [&](const float tmc) {           // raise the reference maximum to cover tmc; rescale O and l
            const float mn = fmaxf(mc[bs], tmc);
            const float alpha = __builtin_amdgcn_exp2f(mc[bs] - mn);
            mc[bs] = mn;
            lrun[bs] *= alpha;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[bs][i][r] *= alpha;
        };
        if (tail) {                         // last tile of a sequence whose length is not a multiple of 64
            // register r of 32-key block kbk holds key kv0 + kbk*32 + 16*(r>>3) + 8*hi + (r&7): valid below `lim`.
            // `lim` passes through an empty asm INSIDE the branch: hipcc otherwise hoists the 32 compares (and
            // if-converts half of the selects) into the code that runs on every tile.
            int lim = S - kv0 - 8 * hi;
            asm volatile("" : "+v"(lim));
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbk * 32 + 16 * (r >> 3) + (r & 7) >= lim) sacc[bs][kbk][r] = -1e30f;
        }
        // Exact reference maximum: on a row's first key tile, and on every tile in exact mode.  A wave-uniform branch
        // outside the pipelined region (the speculative steady state skips it).
        if (need_max) {
            asm volatile("" ::: "memory");                   // keep it a branch (no if-conversion into the hot path)
            const float tmc = tile_max();
            if (__any(tmc > mc[bs] + thr)) rescale_to(tmc);  // thr = 0: the maximum is always exact
            if constexpr (QP) {                              // (exact pass only: c = 1, the pipelined slots take exp2 of the accumulators as they are)
                const float mref = mc[bs];
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[bs][kbk][r] -= mref;
            }
        }
        // 16 x { 1 MFMA (+ the LDS read of the fragment two MFMAs ahead), 7 VALU of three different score pairs }
        ps0 = ps1 = ps2 = ps3 = 0.f;
        const float nm = -mc[bs];
        if constexpr (!QP) pair_fma(0, nm);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mfma_step(m);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PPS; ++u) softmax_slot(m * PPS + u, nm);
            hook(m);                                         // this wave's share of the prefetch DMAs (issued under an MFMA)
            __builtin_amdgcn_sched_barrier(0);
        }
        pair_sum_pack(15, pa0, pa1);
        const float psum = (ps0 + ps1) + (ps2 + ps3);
        // overflow (inf / NaN included): the work item is redone exactly.  fp16 P ends at 65 504: a lane's partial sum below 3e4 bounds each of
        // its P values (the pack would have produced inf otherwise; those MFMAs are discarded with the redo)
        if (__any(!(psum < (F16 ? 3.0e4f : 1e30f)))) ovf = 1;
#if ESME_ATTN_ABL
        lrun[bs] += 1.0f;
#if ESME_ATTN_ABL & 2
        asm volatile("" : "+v"(abl_l));
#endif
#else
        lrun[bs] += psum;
#endif
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // The pass over the key tiles.  It is a loop body so that a workgroup whose speculative pass overflowed (a later key
    // beat a row's first-tile maximum by more than ~2^100, or the scores hold inf / NaN) can redo its work item with the
    // classic online softmax: never taken on real data, but it makes the speculative pass exact, not "fine in practice".
    const int nt = (S + KT - 1) / KT;
    bool exact = !a.spec;
    for (;;) {
        // ---- prologue: K tiles 0..2 and V tiles 0, 1 by LDS-DMA; V of slot 3 zeroed (phase A of iteration 0 multiplies
        // it by P = 0: no NaN / Inf patterns)
    #pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            mc[bb] = -1e30f; lrun[bb] = (QP && !wave_active) ? 1e30f : 0.f;    // (QP: an idle wave must not trip the vanished-sum check)
    #pragma unroll
            for (int i = 0; i < 2; ++i) {
    #pragma unroll
                for (int r = 0; r < 16; ++r) { if (i < DB) oacc[bb][i < DB ? i : 0][r] = 0.f; sacc[bb][i][r] = S0; }
    #pragma unroll
                for (int s = 0; s < 2; ++s) pw[bb][i][s] = u32x4{0u, 0u, 0u, 0u};
            }
        }
        dma_k(0, smem);
        dma_v(0, smem);
        if (nt > 1) { dma_k(1, smem + SLOT); dma_v(1, smem + SLOT); }
        if (nt > 2) dma_k(2, smem + 2 * SLOT);
        // The Q fragments are re-defined by an (empty) asm statement here, while no other load is in flight: hipcc's
        // wait-count pass otherwise carries "the Q loads may still be pending" into the loop header and guards their
        // first uses with s_waitcnt vmcnt(3..0) -- which, in steady state, drains the staging loads issued at the top of
        // the same iteration (an HBM latency per key tile; the kernel ran 1.7x slower on long sequences).
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int ds = 0; ds < DS; ds += 2) asm volatile("" : "+v"(qf[bb][ds]), "+v"(qf[bb][ds + 1]) : : "memory");
        {
            const u32x4 z = {0u, 0u, 0u, 0u};
            char* v3 = smem + 3 * SLOT + K_BYTES;
    #pragma unroll
            for (int i = 0; i < (D * 128) / (NT * 16); ++i) *reinterpret_cast<u32x4*>(v3 + (i * NT + tid) * 16) = z;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the prologue DMAs (hipcc does not know about them)
        __syncthreads();
        if (wave_active) {                      // S^T(b0, tile 0)
    #pragma unroll
            for (int ds = 0; ds < DS; ++ds)
    #pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + kbk * (32 * ROWB) + kfo[ds]);
                    sacc[0][kbk] = mfma_32x32x16<F16>(kf, qf[0][ds], sacc[0][kbk]);
                }
        }
        const bool ragged = (S & (KT - 1)) != 0;
        // One key tile per iteration: phase A: softmax(b0,t) || S^T(b1,t), O^T(b1) += V(t-1) P(b1,t-1);
        // phase B: softmax(b1,t) || S^T(b0,t+1), O^T(b0) += V(t) P(b0,t).  The loop holds ONE instantiation of the phase
        // pair (several variants in branches of the loop make the register allocator spill ~150 VGPRs).
        // iteration t also starts the LDS-DMA of K tile t+3 (K region of slot t-1: last read in phase A of iteration t-1)
        // and of V tile t+2 (V region of slot t-2: last read in phase A of iteration t-1); both are first read in phase B
        // of iteration t+2, two iterations to land.  An LDS-DMA instruction holds its wave for 60-180 cycles until the
        // memory pipeline has taken it, so the pieces are issued one at a time from inside the phases, each right after an
        // MFMA (the matrix pipe keeps running); issued together at the top of the iteration they cost ~600 cycles per tile.
        auto dma_piece = [&](const unsigned int rs_sel, const int tile, char* slot, const int i) {
            if (rs_sel == 0) dma16(krs, (unsigned int)tile * tile_bytes + kg0 + i * kg_step, slot + (i * NW + wave) * 1024);
            else dma16(vrs, (unsigned int)tile * tile_bytes + vg0 + i * kg_step, slot + K_BYTES + (i * NW + wave) * 1024);
        };
        for (int t = 0; t < nt; ++t) {
            const bool pf_k = t + 3 < nt, pf_v = t + 2 < nt;       // block-uniform
            char* kslot = smem + ((t + 3) & 3) * SLOT;
            char* vslot = smem + ((t + 2) & 3) * SLOT;
            if (wave_active) {
                const char* cur = smem + (t & 3) * SLOT;
                const char* prv = smem + ((t + 3) & 3) * SLOT;
                const char* nxt = smem + ((t + 1) & 3) * SLOT;
                const bool tail = t == nt - 1 && ragged;
                const bool need_max = exact || (!QP && t == 0);
                phase(I0{}, I1{}, need_max, tail, cur, prv, t * KT, [&](const int m) {
                    if (KI == 2) { if (m == ESME_ATTN_DMA0 && pf_k) dma_piece(0, t + 3, kslot, 0); if (m == ESME_ATTN_DMA1 && pf_k) dma_piece(0, t + 3, kslot, KI - 1); }
                    else if (m == NM / 2 - 1 && pf_k) dma_piece(0, t + 3, kslot, 0);
                });
                phase(I1{}, I0{}, need_max, tail, nxt, cur, t * KT, [&](const int m) {
                    if (KI == 2) { if (m == ESME_ATTN_DMA0 && pf_v) dma_piece(1, t + 2, vslot, 0); if (m == ESME_ATTN_DMA1 && pf_v) dma_piece(1, t + 2, vslot, KI - 1); }
                    else if (m == NM / 2 - 1 && pf_v) dma_piece(1, t + 2, vslot, 0);
                });
            } else {
                if (pf_k) dma_k(t + 3, kslot);
                if (pf_v) dma_v(t + 2, vslot);
            }
            // End of iteration t.  NOT __syncthreads(): with an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front of
            // the barrier, i.e. every key tile would wait for the prefetch issued at its own top.  What must have landed
            // here are the tiles first read in iteration t+1 (K tile t+2, V tile t+1: issued in iteration t-1); the DMAs this
            // iteration issued may stay in flight.
            if (t + 3 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * KI) : "memory");
            else if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(KI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (wave_active) {                          // drain: O^T(b1) += V(nt-1) P(b1, nt-1)
            const char* Vs = smem + ((nt - 1) & 3) * SLOT;
    #pragma unroll
            for (int ks = 0; ks < 4; ++ks)
    #pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const bf16x8 vf = vfrag(Vs, db, ks);
                    oacc[1][db] = mfma_32x32x16<F16>(vf, __builtin_bit_cast(bf16x8, pw[1][ks >> 1][ks & 1]),
                                                                          oacc[1][db]);
                }
        }
        if constexpr (QP) {
            // no reference maximum: a row whose every score sits below ~-100 (log2 units) has lost its sum -- redo it exactly.
            // (The ROW's sum, i.e. both key halves: lane and lane ^ 32 combined -- a lane whose half of the keys is entirely masked, as in
            // sequences of <= 8 residues, holds 0 by itself and used to send such sequences through the loop twice.)
            if (!exact) {
                // F16: a row sum below S * 2^-14 means the row's P values average below fp16's smallest NORMAL number -- the terms that carry the row's weight
                // would sit in the subnormals (fewer than 11 bits; the row sum itself is taken from the unrounded values, so the loss does not cancel).  Above it the
                // subnormal terms together are worth at most 2^-11 of the row (S terms x 2^-25 absolute against a sum >= S * 2^-14): one fp16 rounding.
                const float vanish = F16 ? (float)S * 6.103515625e-5f : 1e-30f;
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lrun[bb]), __float_as_uint(lrun[bb]), false, false);
                    if (__any(!(__uint_as_float(sw[0]) + __uint_as_float(sw[1]) > vanish))) ovf = 1;      // (idle waves carry lrun = 1e30)
                }
            }
        }
        if (exact || !__syncthreads_or(ovf)) break;
        exact = true;
        ovf = 0;
    }
    if (!wave_active) return;

    // ---- epilogue: normalise, transpose through a wave-private LDS slab (a slot no wave reads any more: every
    // wave passed the loop's last barrier, only V of tile nt-1 is still in use), store whole 128-B rows
    constexpr int OCH = D / 8;                              // 16-B chunks per output row
    char* slab = smem + ((nt + (wave >> 2)) & 3) * SLOT + (wave & 3) * (32 * ROWB);
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lrun[bb]), __float_as_uint(lrun[bb]), false, false);
        const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
        if (bb) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk = {pack16<F16>(oacc[bb][db][4 * g] * inv, oacc[bb][db][4 * g + 1] * inv),
                            pack16<F16>(oacc[bb][db][4 * g + 2] * inv, oacc[bb][db][4 * g + 3] * inv)};
                ESME_LDS_CHECK(slab + l31 * ROWB + (((db * 4 + g) ^ (l31 & (OCH - 1))) << 4) + hi * 8, 8, smem, 4 * SLOT);
                *reinterpret_cast<u32x2*>(slab + l31 * ROWB + (((db * 4 + g) ^ (l31 & (OCH - 1))) << 4) + hi * 8) = pk;
            }
        __builtin_amdgcn_wave_barrier();
        const int rbase = wrow0 + bb * 32;
#pragma unroll
        for (int it = 0; it < 32 / (64 / OCH); ++it) {
            const int r = it * (64 / OCH) + lane / OCH, ch = lane % OCH;
            ESME_LDS_CHECK(slab + r * ROWB + ((ch ^ (r & (OCH - 1))) << 4), 16, smem, 4 * SLOT);
            const u32x4 v = *reinterpret_cast<const u32x4*>(slab + r * ROWB + ((ch ^ (r & (OCH - 1))) << 4));
            if (rbase + r < S) *reinterpret_cast<u32x4*>(a.o + (int64_t)(s0 + rbase + r) * a.ldo + h * D + ch * 8) = v;
        }
    }
}


// =============================================================================================
// Head dims 64 / 32, ONE 32-row query block per wave, software-pipelined over KEY TILES (round 6): "attn_sb".
//
// attn_pp64_kernel hides a block's softmax under the MFMAs of the wave's OTHER block; that takes two blocks' accumulators and Q fragments
// (233 of 256 registers at head dim 64), which is why the q/k-pair form of precision 'half' (three score passes: + the lo fragments of q, 32
// registers) ran on the first-generation, non-pipelined structure at 14 % of the matrix peak.  Here a wave owns ONE
// block and the pipeline runs along the key axis instead: in phase t
//     VALU:  softmax of tile t's scores (parity buffer sacc[t & 1])  ->  P(t) (pw[t & 1]), row sums
//     MFMA:  O^T += V(t-1)^T P(t-1)^T  (P of the previous phase)   then   S^T(t+1) = K(t+1) Q^T  into the other parity buffer
// -- the same three independent streams as the ping-pong kernel with half the accumulators: 64 (scores) + 32 (O^T) + 16 (Q) [+ 16 (Q lo)] + 32 (P)
// registers.  With q / k as pairs (QKP) a phase issues 8 + 3 x 8 = 32 MFMAs against the same 16 score pairs: the loop turns from VALU-bound to
// MFMA-bound, i.e. the two extra score passes cost matrix-pipe time the plain kernel leaves idle (52 % busy) rather than a second trip.
// Online softmax with a DEFERRED rescale: the MFMAs of phase t add P(t-1), which was formed against the reference maximum of phase t-1, so a maximum
// raised in phase t rescales the row sum at once but O^T only after the phase's MFMAs (before P(t) is added in phase t + 1).
// LDS: ring of THREE slots (K tile [+ K lo tile] + V tile); phase t reads K(t+1) and V(t-1), prefetches K(t+3) into K(t)'s slot and V(t+1) into
// V(t-2)'s (both last read in phase t-1; two phases to land, counted vmcnt + one raw barrier per phase as in attn_pp64_kernel).  48 KB (72 KB with
// K lo) per workgroup of 4 waves = 128 query rows: two workgroups per CU.
// QP / F16 / speculative pass / redo: exactly attn_pp64_kernel's (see there); QKP runs with exact maxima (a.spec = 0): with scores in the
// hundreds a later key beats the first tile's maximum by more than fp16's range on most rows, and the speculative pass would be redone anyway.
template <int D, bool F16, bool QKP, bool QP>
__global__ __launch_bounds__(256, 2) void attn_sb_kernel(const AttnSplitArgs sa) {
    static_assert(!QKP || F16, "q/k pairs: the fp16 form");
    static_assert(!(F16 && QP), "fp16 P needs a reference maximum");
    static_assert(D == 64 || D == 32, "head dims 64 and 32");
    const AttnArgs& a = sa.a;
    constexpr int NW = 4, NT = 256, DS = D / 16, DB = D / 32;
    constexpr int ROWB = D * 2, K_BYTES = KT * D * 2, KP = QKP ? 2 : 1;
    constexpr int V_OFF = KP * K_BYTES, SLOT = V_OFF + D * 128, NS = 3, ROWS = NW * 32;
    constexpr int NPV = 4 * DB, NQK1 = 2 * DS, NQK = NQK1 * (QKP ? 3 : 1), NM = NPV + NQK;      // MFMAs of a phase
    extern __shared__ __attribute__((aligned(16))) char smem_sb[];
    char* smem = smem_sb;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned int xcd = blockIdx.x & 7u, bi = blockIdx.x >> 3;          // (block id -> work item: as attn_pp64_kernel)
    const int qt = (int)(bi % (unsigned int)a.nqt);
    const unsigned int hb = (bi / (unsigned int)a.nqt) * 8u + xcd;
    if (hb >= (unsigned int)a.nhb) return;
    const int h = (int)(hb % (unsigned int)a.H), bi_seq = (int)(hb / (unsigned int)a.H);
    const int b = a.order ? a.order[bi_seq] : bi_seq;
    const int s0 = a.cu[b], S = a.cu[b + 1] - s0;
    const int q0 = qt * ROWS;
    if (q0 >= S) return;

    const unsigned int ld = (unsigned int)a.ld;
    const u16* qb = a.q + (int64_t)s0 * a.ld + h * D;
    const unsigned int kv_bytes = ((unsigned int)(S - 1) * ld + D) * 2u;
    auto make_rsrc = [&](const u16* p) -> u32x4 {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        return u32x4{(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v),
                     (unsigned int)__builtin_amdgcn_readfirstlane((int)((unsigned int)(v >> 32) & 0xffffu)),
                     (unsigned int)__builtin_amdgcn_readfirstlane((int)kv_bytes), 0x00020000u};
    };
    const u32x4 krs = make_rsrc(a.k + (int64_t)s0 * a.ld + h * D);
    const u32x4 klrs = make_rsrc(a.k + (QKP ? sa.lo_in : 0) + (int64_t)s0 * a.ld + h * D);
    const u32x4 vrs = make_rsrc(a.v + (int64_t)s0 * a.ld + h * D);
    auto dma16 = [&](const u32x4 rs, const unsigned int voff, const char* dst) {
        ESME_LDS_CHECK(dst, 1024, smem, NS * SLOT);
        const unsigned int d = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(uintptr_t)dst);
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(d), "s"(rs) : "memory");
    };

    // ---- Q fragments of the wave's block (B operand of S^T): lane (q = l31, hi) holds Q[q][ds*16 + hi*8 ..]; QKP: the lo halves too
    bf16x8 qf[DS], qfl[QKP ? DS : 1];
    const int wrow0 = q0 + wave * 32;
    const bool wave_active = wrow0 < S;
    {
        const int qr = wrow0 + l31;
        const unsigned int qc = qr < S ? qr : S - 1;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
            qf[ds] = *reinterpret_cast<const bf16x8*>(qb + (qc * ld + ds * 16 + hi * 8));
            if constexpr (QKP) qfl[ds] = *reinterpret_cast<const bf16x8*>(qb + sa.lo_in + (qc * ld + ds * 16 + hi * 8));
        }
    }

    // ---- staging (as attn_pp64_kernel: LDS-DMA, source-address swizzle, descriptor bounds = zeros past the sequence end)
    constexpr int KI = K_BYTES / (NW * 1024);                    // DMA instructions per wave per K (or V) tile part (2 or 1)
    constexpr int CPR = D / 8;
    const unsigned int tile_bytes = (unsigned int)KT * ld * 2u;
    unsigned int kg0, vg0;
    {
        const int r = wave * (64 / CPR) + lane / CPR, pch = lane % CPR;
        kg0 = ((unsigned int)r * ld + ((pch ^ kswz<D>(r)) * 8)) * 2u;
        vg0 = ((unsigned int)r * ld + ((D == 64 ? (pch ^ (((r >> 1) & 1) << 2)) : pch) * 8)) * 2u;
    }
    const unsigned int kg_step = (unsigned int)(NW * (64 / CPR)) * ld * 2u;
    // piece p of a tile's prefetch: [0, KI) K hi, [KI, KP*KI) K lo, then KI pieces of V
    auto dma_k_piece = [&](const int tile, char* slot, const int p) {
        const int part = p / KI, i = p % KI;
        dma16(part ? klrs : krs, (unsigned int)tile * tile_bytes + kg0 + i * kg_step, slot + part * K_BYTES + (i * NW + wave) * 1024);
    };
    auto dma_v_piece = [&](const int tile, char* slot, const int i) {
        dma16(vrs, (unsigned int)tile * tile_bytes + vg0 + i * kg_step, slot + V_OFF + (i * NW + wave) * 1024);
    };
    auto dma_k = [&](const int tile, char* slot) {
#pragma unroll
        for (int p = 0; p < KP * KI; ++p) dma_k_piece(tile, slot, p);
    };
    auto dma_v = [&](const int tile, char* slot) {
#pragma unroll
        for (int i = 0; i < KI; ++i) dma_v_piece(tile, slot, i);
    };

    const int krow_perm = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | (l31 & 16);
    int kfo[DS];
#pragma unroll
    for (int i = 0; i < DS; ++i) kfo[i] = krow_perm * ROWB + (((i * 2 + hi) ^ kswz<D>(krow_perm)) << 4);
    int vb[DB];
    {
        const int j = (lane & 15) >> 2, p = lane & 3, gsel = (lane >> 4) & 1;
#pragma unroll
        for (int db = 0; db < DB; ++db) vb[db] = V_OFF + (hi * 8 + j) * ROWB + (D == 64 ? ((db ^ (j >> 1)) * 64) : 0) + gsel * 32 + p * 8;
    }
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    auto vfrag = [&](const char* Vs, const int db, const int ks) -> bf16x8 {
        typedef __attribute__((address_space(3))) s16x4* ltr_t;
        const char* p = Vs + vb[db] + ks * (16 * ROWB);
        ESME_LDS_CHECK(p, 8, smem, NS * SLOT); ESME_LDS_CHECK(p + 4 * ROWB, 8, smem, NS * SLOT);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(p));
        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(p + 4 * ROWB));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 oacc[DB], sacc[2][2];       // sacc[parity of the key tile][32-key block]
    u32x4 pw[2][2][2];                 // P of tile parity p as packed 16-bit values: [p][32-key block][16-key step]
    float mc, lrun;
    const float c = a.scale_log2, thr = a.thr;
    int ovf = 0;

    // MFMA m of a phase.  m < NPV: O^T += V^T P^T (db-minor: consecutive MFMAs alternate between the two accumulators); then the score passes,
    // 32-key-block-minor.  QKP pass order: Kh Qh^T (clears the accumulator), Kl Qh^T, Kh Ql^T.
    auto frag = [&](const int m, const char* Ks, const char* Vs) -> bf16x8 {
        if (m < NPV) return vfrag(Vs, m % DB, m / DB);
        const int j = (m - NPV) % NQK1, pass = (m - NPV) / NQK1;
        const char* p = Ks + (pass == 1 ? K_BYTES : 0) + (j & 1) * (32 * ROWB) + kfo[j >> 1];
        ESME_LDS_CHECK(p, 16, smem, NS * SLOT);
        return *reinterpret_cast<const bf16x8*>(p);
    };

    // One phase: softmax of the scores in sacc[P] (key tile at kv0), interleaved with the phase's MFMAs (P(t-1) V(t-1) into O^T, K(t+1) Q^T into
    // sacc[P ^ 1]).  Returns nothing; `alpha` != 1 (wave-uniform flag `resc`) is applied to O^T after the MFMAs.
    auto phase = [&](auto P_, const bool need_max, const bool tail, const char* Ks, const char* Vs, const int kv0, auto&& hook) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value, PN = P ^ 1;
        bf16x8 fr[3];
        fr[0] = frag(0, Ks, Vs);
        fr[1] = frag(1, Ks, Vs);
        auto mfma_step = [&](const int m) {
            if (m + 2 < NM) fr[(m + 2) % 3] = frag(m + 2, Ks, Vs);
            if (m >= NPV) {
                const int j = (m - NPV) % NQK1, pass = (m - NPV) / NQK1, kbk = j & 1, ds = j >> 1;
                const bf16x8 qop = (QKP && pass == 2) ? qfl[QKP ? ds : 0] : qf[ds];
                if (pass == 0 && ds == 0) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    sacc[PN][kbk] = mfma_32x32x16<F16>(fr[m % 3], qop, z);
                } else {
                    sacc[PN][kbk] = mfma_32x32x16<F16>(fr[m % 3], qop, sacc[PN][kbk]);
                }
            } else {
                const int db = m % DB, ks = m / DB;
                oacc[db] = mfma_32x32x16<F16>(fr[m % 3], __builtin_bit_cast(bf16x8, pw[PN][ks >> 1][ks & 1]), oacc[db]);
            }
        };
        float ps0, ps1, ps2, ps3;
        float xa0, xa1, pa0 = 0.f, pa1 = 0.f;
        auto pair_fma = [&](const int p, const float nm) {
            const int kbk = p >> 3, r = (2 * p) & 15;
            xa0 = fmaf(sacc[P][kbk][r], c, nm);
            xa1 = fmaf(sacc[P][kbk][r + 1], c, nm);
        };
        auto pair_sum_pack = [&](const int p, const float q0_, const float q1_) {
            const int kbk = p >> 3, r = (2 * p) & 15;
            pw[P][kbk][r >> 3][(r & 7) >> 1] = pack16<F16>(q0_, q1_);
            if (p & 1) { ps2 += q0_; ps3 += q1_; } else { ps0 += q0_; ps1 += q1_; }
        };
        auto softmax_slot = [&](const int m, const float nm) {      // pair step m = 0..15
            const float q0_ = pa0, q1_ = pa1;
            if constexpr (QP) {
                const int kb2 = m >> 3, r2 = (2 * m) & 15;
                pa0 = __builtin_amdgcn_exp2f(sacc[P][kb2][r2]);
                pa1 = __builtin_amdgcn_exp2f(sacc[P][kb2][r2 + 1]);
                if (m >= 1) pair_sum_pack(m - 1, q0_, q1_);
                const int pk = m >= 1 ? m - 1 : 0, kbk = pk >> 3, r = (2 * pk) & 15;
                asm volatile("" : "+v"(pw[P][kbk][r >> 3]), "+v"(ps0), "+v"(ps1), "+v"(ps2), "+v"(ps3), "+v"(pa0), "+v"(pa1));
            } else {
                const float x0 = xa0, x1 = xa1;
                if (m + 1 < 16) pair_fma(m + 1, nm);
                pa0 = __builtin_amdgcn_exp2f(x0);
                pa1 = __builtin_amdgcn_exp2f(x1);
                if (m >= 1) pair_sum_pack(m - 1, q0_, q1_);
                const int pk = m >= 1 ? m - 1 : 0, kbk = pk >> 3, r = (2 * pk) & 15;
                asm volatile("" : "+v"(pw[P][kbk][r >> 3]), "+v"(ps0), "+v"(ps1), "+v"(ps2), "+v"(ps3), "+v"(xa0), "+v"(xa1), "+v"(pa0), "+v"(pa1));
            }
        };
        auto tile_max = [&]() -> float {
            float tmax = sacc[P][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sacc[P][0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[P][1][r]);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * c;
        };
        if (tail) {
            int lim = S - kv0 - 8 * hi;
            asm volatile("" : "+v"(lim));
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbk * 32 + 16 * (r >> 3) + (r & 7) >= lim) sacc[P][kbk][r] = -1e30f;
        }
        float alpha = 1.0f;
        bool resc = false;
        if (need_max) {
            asm volatile("" ::: "memory");
            const float tmc = tile_max();
            if (__any(tmc > mc + thr)) {                      // raise the reference maximum: the row sum now, O^T after this phase's MFMAs
                const float mn = fmaxf(mc, tmc);
                alpha = __builtin_amdgcn_exp2f(mc - mn);
                mc = mn;
                lrun *= alpha;
                resc = true;
            }
            if constexpr (QP) {
                const float mref = mc;
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[P][kbk][r] -= mref;
            }
        }
        ps0 = ps1 = ps2 = ps3 = 0.f;
        const float nm = -mc;
        if constexpr (!QP) pair_fma(0, nm);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mfma_step(m);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NM >= 16) { if (m % (NM / 16) == 0) softmax_slot(m / (NM / 16), nm); }
            else {
#pragma unroll
                for (int u = 0; u < 16 / NM; ++u) softmax_slot(m * (16 / NM) + u, nm);
            }
            hook(m);
            __builtin_amdgcn_sched_barrier(0);
        }
        pair_sum_pack(15, pa0, pa1);
        const float psum = (ps0 + ps1) + (ps2 + ps3);
        if (__any(!(psum < (F16 ? 3.0e4f : 1e30f)))) ovf = 1;
        lrun += psum;
        if (resc) {                                           // (wave-uniform) the deferred rescale of O^T
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    const int nt = (S + KT - 1) / KT;
    bool exact = !a.spec;
    constexpr int NPK = KP * KI, NPIECE = NPK + KI;            // prefetch DMA instructions of a phase per wave: K(t+3) parts, V(t+1)
    for (;;) {
        mc = -1e30f; lrun = (QP && !wave_active) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { if (i < DB) oacc[i < DB ? i : 0][r] = 0.f; sacc[0][i][r] = 0.f; sacc[1][i][r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 2; ++s) { pw[0][i][s] = u32x4{0u, 0u, 0u, 0u}; pw[1][i][s] = u32x4{0u, 0u, 0u, 0u}; }
        }
        // ---- prologue: K tiles 0..2 and V tile 0 by LDS-DMA; V of slot 2 (= "tile -1") zeroed: phase 0 multiplies it by P = 0
        dma_k(0, smem);
        dma_v(0, smem);
        if (nt > 1) dma_k(1, smem + SLOT);
        if (nt > 2) dma_k(2, smem + 2 * SLOT);
#pragma unroll
        for (int ds = 0; ds < DS; ds += 2) asm volatile("" : "+v"(qf[ds]), "+v"(qf[ds + 1]) : : "memory");
        if constexpr (QKP) {
#pragma unroll
            for (int ds = 0; ds < DS; ds += 2) asm volatile("" : "+v"(qfl[ds]), "+v"(qfl[ds + 1]) : : "memory");
        }
        {
            const u32x4 z = {0u, 0u, 0u, 0u};
            char* v2 = smem + 2 * SLOT + V_OFF;
#pragma unroll
            for (int i = 0; i < (D * 128) / (NT * 16); ++i) *reinterpret_cast<u32x4*>(v2 + (i * NT + tid) * 16) = z;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wave_active) {                      // S^T(tile 0) into sacc[0]
#pragma unroll
            for (int pass = 0; pass < (QKP ? 3 : 1); ++pass)
#pragma unroll
                for (int ds = 0; ds < DS; ++ds)
#pragma unroll
                    for (int kbk = 0; kbk < 2; ++kbk) {
                        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + (pass == 1 ? K_BYTES : 0) + kbk * (32 * ROWB) + kfo[ds]);
                        sacc[0][kbk] = mfma_32x32x16<F16>(kf, (QKP && pass == 2) ? qfl[QKP ? ds : 0] : qf[ds], sacc[0][kbk]);
                    }
        }
        const bool ragged = (S & (KT - 1)) != 0;
        // phase t: reads K(t+1) [slot (t+1) % 3] and V(t-1) [slot (t+2) % 3]; prefetches K(t+3) into slot t % 3 and V(t+1) into slot (t+1) % 3's V
        // region (V(t-2)'s).  At its end: K(t+2) and V(t) (issued in phase t-1) must have landed; this phase's own DMAs may stay in flight.
        auto step = [&](auto P_, const int t) __attribute__((always_inline)) {
            const bool pf_k = t + 3 < nt, pf_v = t + 1 < nt;         // block-uniform
            char* kslot = smem + (t % NS) * SLOT;
            char* vslot = smem + ((t + 1) % NS) * SLOT;
            if (wave_active) {
                const char* Ks = smem + ((t + 1) % NS) * SLOT;
                const char* Vs = smem + ((t + 2) % NS) * SLOT;
                const bool tail = t == nt - 1 && ragged;
                const bool need_max = exact || (!QP && t == 0);
                phase(P_, need_max, tail, Ks, Vs, t * KT, [&](const int m) {
#pragma unroll
                    for (int p = 0; p < NPIECE; ++p) {
                        if (m == ((2 * p + 1) * NM) / (2 * NPIECE)) {        // the pieces spread over the phase, each behind an MFMA
                            if (p < NPK) { if (pf_k) dma_k_piece(t + 3, kslot, p); }
                            else if (pf_v) dma_v_piece(t + 1, vslot, p - NPK);
                        }
                    }
                });
            } else {
                if (pf_k) dma_k(t + 3, kslot);
                if (pf_v) dma_v(t + 1, vslot);
            }
            if (pf_k) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NPIECE) : "memory");
            else if (pf_v) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(KI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        // (V(0) was drained with the prologue; V(1) is issued in phase 0 and first read in phase 2)
        for (int t = 0; t < nt; t += 2) {
            step(I0{}, t);
            if (t + 1 < nt) step(I1{}, t + 1);
        }
        if (wave_active) {                          // drain: O^T += V(nt-1) P(nt-1)
            const char* Vs = smem + ((nt - 1) % NS) * SLOT;
            auto drain = [&](auto P_) __attribute__((always_inline)) {
                constexpr int P = decltype(P_)::value;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int db = 0; db < DB; ++db)
                        oacc[db] = mfma_32x32x16<F16>(vfrag(Vs, db, ks), __builtin_bit_cast(bf16x8, pw[P][ks >> 1][ks & 1]), oacc[db]);
            };
            if ((nt - 1) & 1) drain(I1{}); else drain(I0{});
        }
        if constexpr (QP) {
            if (!exact) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lrun), __float_as_uint(lrun), false, false);
                if (__any(!(__uint_as_float(sw[0]) + __uint_as_float(sw[1]) > 1e-30f))) ovf = 1;
            }
        }
        if (exact || !__syncthreads_or(ovf)) break;
        exact = true;
        ovf = 0;
    }
    if (!wave_active) return;

    // ---- epilogue: normalise, transpose through a wave-private slab in the slot no wave reads any more (tile nt's), whole rows out
    constexpr int OCH = D / 8;
    char* slab = smem + (nt % NS) * SLOT + wave * (32 * ROWB);
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lrun), __float_as_uint(lrun), false, false);
        const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk = {pack16<F16>(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv),
                            pack16<F16>(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv)};
                ESME_LDS_CHECK(slab + l31 * ROWB + (((db * 4 + g) ^ (l31 & (OCH - 1))) << 4) + hi * 8, 8, smem, NS * SLOT);
                *reinterpret_cast<u32x2*>(slab + l31 * ROWB + (((db * 4 + g) ^ (l31 & (OCH - 1))) << 4) + hi * 8) = pk;
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / (64 / OCH); ++it) {
            const int r = it * (64 / OCH) + lane / OCH, ch = lane % OCH;
            const u32x4 v = *reinterpret_cast<const u32x4*>(slab + r * ROWB + ((ch ^ (r & (OCH - 1))) << 4));
            if (wrow0 + r < S) *reinterpret_cast<u32x4*>(a.o + (int64_t)(s0 + wrow0 + r) * a.ldo + h * D + ch * 8) = v;
        }
    }
}

#ifdef ESME_ATTN_W4          // lab build only (tools/lab/build_alt.sh attn.hip ... with -DESME_ATTN_W4): measured slower, see the note below
// =============================================================================================
// Head dim 64, ONE wave per SIMD (round 4): 4 waves per workgroup, each wave owns FOUR 32-row query blocks (128 rows; 512 per
// workgroup) and the whole 512-entry register file of its SIMD (O^T accumulators and the Q fragments in the accumulator half).
// Same math, LDS images and data path as attn_pp64_kernel (K / V tiles by counted LDS-DMA into a ring of four slots, V^T fragments by
// ds_read_b64_tr_b16, speculative softmax on pre-scaled q); what changes is the amount of work per fragment and per barrier:
//   * every K / V^T fragment is read ONCE per key tile into registers and feeds FOUR MFMAs (one per query block): 24 LDS
//     instructions per 64 MFMAs instead of ~1.5 per MFMA, one barrier per 64 MFMAs instead of per 32;
//   * the software pipeline rotates over the four blocks: phase (b, t) runs softmax(b, t) on the VALU while the matrix pipe does
//     O^T(b-1) += V^T P(b-1)^T and S^T(b+1) = K Q_{b+1}^T -- 16 x { 1 MFMA, 1 score pair: 2 v_exp + 2 v_add + 1 v_cvt_pk };
//     the K fragment set is replaced in place (tile t+1) behind the MFMAs of phase (2, t), the V^T set behind those of (0, t+1).
// MEASURED (profiles/r04_attn_w4_lab.txt; bit-identical to attn_pp64_kernel<4, true> on every batch): 237 vs 183 us at S = 500, 246 vs
// 194 at S = 1 002, 663 vs 546 at S = 2 000, 504 vs 355 on the proteome-like batch -- 20-40 % SLOWER.  Ablations at S = 2 000: without
// the in-loop LDS-DMA 607 us, without the fragment reloads 615: the core {1 MFMA, 2 v_exp, 2 v_add, 1 v_cvt_pk} stream of ONE wave runs
// ~64 cycles per MFMA, which is what tools/lab/mfma_issue_probe.hip predicts (profiles/r02_mfma_issue_probe.txt: this mix at F = 5 costs
// 49.7 cycles per MFMA with one wave per SIMD and 38.9 with two -- a second wave hides the VALU issue of the first behind its own MFMA;
// at head dim 128, where the CDNA4 guide's one-wave kernel reaches 50-56 %, every MFMA carries half the softmax work).  Kept out of the
// shipped library; not a candidate.
// q must arrive pre-multiplied by softmax_scale * log2(e) (esme_attn_opts_t.q_prescaled; the QKV epilogue / the ESM-C q/k pass do
// it): P = exp2(score) with no reference maximum; a row sum that overflows or vanishes sends the work item through the classic
// online softmax (the same code with need_max), exactly as in attn_pp64_kernel<NW, true>.
// The MFMAs are inline asm with the accumulator classes pinned -- O^T and Q in the AGPR half ("a"), scores in arch VGPRs ("v":
// the softmax reads them; a VALU instruction cannot read an AGPR) -- hipcc's own placement of the builtin copies accumulators
// between the halves.  Hazards: an MFMA result is read by the VALU at the earliest two MFMAs (>= 64 cycles) after the MFMA that
// wrote it (kbk-minor order of the S^T steps; the need_max path pads with s_nop); an accumulate chain needs no wait state.
#ifndef ESME_W4_ABL
#define ESME_W4_ABL 0           // timing ablations (wrong results): 1 = no softmax VALU, 2 = no LDS-DMA inside the loop, 4 = no fragment reloads
#endif
#define ESME_MFMA32_VA(ACC, AF, BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(AF), "a"(BF))
#define ESME_MFMA32_VA0(ACC, AF, BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(ACC) : "v"(AF), "a"(BF))
#define ESME_MFMA32_AV(ACC, AF, BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(AF), "v"(BF))
__global__ __launch_bounds__(256, 1) void attn_w4_kernel(const AttnArgs a) {
    constexpr int D = 64, DS = 4, NW = 4, NT = 256, QB = 4;
    constexpr int K_BYTES = KT * D * 2;          // 8 KB
    constexpr int SLOT = K_BYTES + D * 128;
    constexpr int ROWS = NW * QB * 32;           // 512
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned int xcd = blockIdx.x & 7u, bi = blockIdx.x >> 3;
    const int qt = (int)(bi % (unsigned int)a.nqt);
    const unsigned int hb = (bi / (unsigned int)a.nqt) * 8u + xcd;
    if (hb >= (unsigned int)a.nhb) return;
    const int h = (int)(hb % (unsigned int)a.H), bi_seq = (int)(hb / (unsigned int)a.H);
    const int b = a.order ? a.order[bi_seq] : bi_seq;
    const int s0 = a.cu[b], S = a.cu[b + 1] - s0;
    const int q0 = qt * ROWS;
    if (q0 >= S) return;

    const unsigned int ld = (unsigned int)a.ld;
    const u16* qb = a.q + (int64_t)s0 * a.ld + h * D;
    const unsigned int kv_bytes = ((unsigned int)(S - 1) * ld + D) * 2u;
    auto make_rsrc = [&](const u16* p) -> u32x4 {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        return u32x4{(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v),
                     (unsigned int)__builtin_amdgcn_readfirstlane((int)((unsigned int)(v >> 32) & 0xffffu)),
                     (unsigned int)__builtin_amdgcn_readfirstlane((int)kv_bytes), 0x00020000u};
    };
    const u32x4 krs = make_rsrc(a.k + (int64_t)s0 * a.ld + h * D);
    const u32x4 vrs = make_rsrc(a.v + (int64_t)s0 * a.ld + h * D);
    auto dma16 = [&](const u32x4 rs, const unsigned int voff, const char* dst) {
        const unsigned int d = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(uintptr_t)dst);
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(d), "s"(rs) : "memory");
    };

    // ---- Q fragments of the wave's four q-blocks (B operand of S^T), kept in the accumulator half
    bf16x8 qf[QB][DS];
    const int wrow0 = q0 + wave * (QB * 32);
    const bool wave_active = wrow0 < S;
#pragma unroll
    for (int bb = 0; bb < QB; ++bb) {
        const int qr = wrow0 + bb * 32 + l31;
        const unsigned int qc = qr < S ? qr : S - 1;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds)
            qf[bb][ds] = *reinterpret_cast<const bf16x8*>(qb + (qc * ld + ds * 16 + hi * 8));
    }

    constexpr int KI = 8 / NW;                                   // 2 DMA instructions per wave per K (or V) tile
    const unsigned int tile_bytes = (unsigned int)KT * ld * 2u;
    unsigned int kg0, vg0;
    {
        const int r = wave * 8 + (lane >> 3), pch = lane & 7;
        kg0 = ((unsigned int)r * ld + ((pch ^ kswz<D>(r)) * 8)) * 2u;
        vg0 = ((unsigned int)r * ld + ((pch ^ (((r >> 1) & 1) << 2)) * 8)) * 2u;
    }
    const unsigned int kg_step = (unsigned int)(NW * 8) * ld * 2u;
    auto dma_piece = [&](const unsigned int rs_sel, const int tile, char* slot, const int i) {
        if (rs_sel == 0) dma16(krs, (unsigned int)tile * tile_bytes + kg0 + i * kg_step, slot + (i * NW + wave) * 1024);
        else dma16(vrs, (unsigned int)tile * tile_bytes + vg0 + i * kg_step, slot + K_BYTES + (i * NW + wave) * 1024);
    };
    auto dma_k = [&](int tile, char* slot) {
#pragma unroll
        for (int i = 0; i < KI; ++i) dma_piece(0, tile, slot, i);
    };
    auto dma_v = [&](int tile, char* slot) {
#pragma unroll
        for (int i = 0; i < KI; ++i) dma_piece(1, tile, slot, i);
    };

    const int krow_perm = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | (l31 & 16);
    int kfo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kfo[i] = krow_perm * 128 + (((i * 2 + hi) ^ ((krow_perm >> 1) & 7)) << 4);
    int vb[2];
    {
        const int j = (lane & 15) >> 2, p = lane & 3, gsel = (lane >> 4) & 1;
#pragma unroll
        for (int db = 0; db < 2; ++db) vb[db] = K_BYTES + (hi * 8 + j) * 128 + ((db ^ (j >> 1)) * 64) + gsel * 32 + p * 8;
    }
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    auto vfrag = [&](const char* Vs, const int db, const int ks) -> bf16x8 {
        typedef __attribute__((address_space(3))) s16x4* ltr_t;
        const char* p = Vs + vb[db] + ks * 2048;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(p));
        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(p + 512));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto kfrag = [&](const char* Ks, const int kbk, const int ds) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(Ks + kbk * 4096 + kfo[ds]);
    };

    f32x16 oacc[QB][2];                // O^T of the four blocks: accumulator half
    f32x16 sacc[2][2];                 // scores: set b & 1 of block b, [32-key block]
    u32x4 pw[2][2][2];                 // P as packed bf16: set b & 1, [32-key block][16-key step]
    bf16x8 kfr[2][4], vfr[2][4];       // the K fragment set [kbk][ds] and the V^T set [db][ks] of the tiles in use
    float mc[QB], lrun[QB];
    const float thr = a.thr;
    int ovf = 0;

    using std::integral_constant;
    // One phase: softmax of block B on its finished scores, interleaved with the 16 MFMAs O^T(B-1) += V^T P(B-1)^T (m < 8) and
    // S^T(B+1) = K Q^T (m >= 8).  RK / RV: this is the last phase that uses the K / V^T fragment set -- each fragment is
    // replaced in place (from Kn / Vn, the next tile's slots) right behind the MFMA that used it last.
    auto phase = [&](auto B_, const bool need_max, const bool tail, const int kv0, auto RK_, const char* Kn, auto RV_, const char* Vn, auto&& hook) __attribute__((always_inline)) {
        constexpr int B = decltype(B_)::value, bs = B, bp = (B + 3) & 3, bq = (B + 1) & 3;
        constexpr bool RK = decltype(RK_)::value, RV = decltype(RV_)::value;
        constexpr int ss = bs & 1, sq = bq & 1, sp = bp & 1;          // register sets
        auto mfma_step = [&](const int m) {
            if (m < 8) {
                const int db = m & 1, ks = m >> 1;
                ESME_MFMA32_AV(oacc[bp][db], vfr[db][ks], pw[sp][ks >> 1][ks & 1]);
                if constexpr (RV && !(ESME_W4_ABL & 4)) vfr[db][ks] = vfrag(Vn, db, ks);
            } else {
                const int j = m - 8, kbk = j & 1, ds = j >> 1;
                if (ds == 0) ESME_MFMA32_VA0(sacc[sq][kbk], kfr[kbk][ds], qf[bq][ds]);
                else ESME_MFMA32_VA(sacc[sq][kbk], kfr[kbk][ds], qf[bq][ds]);
                if constexpr (RK && !(ESME_W4_ABL & 4)) kfr[kbk][ds] = kfrag(Kn, kbk, ds);
            }
        };
        float ps0, ps1, ps2, ps3;
        float pa0 = 0.f, pa1 = 0.f;
        auto pair_sum_pack = [&](const int p, const float q0_, const float q1_) {
            const int kbk = p >> 3, r = (2 * p) & 15;
            pw[ss][kbk][r >> 3][(r & 7) >> 1] = pack_bf16(q0_, q1_);
            if (p & 1) { ps2 += q0_; ps3 += q1_; } else { ps0 += q0_; ps1 += q1_; }
        };
        auto softmax_slot = [&](const int m) {
            const float q0_ = pa0, q1_ = pa1;
            const int kb2 = m >> 3, r2 = (2 * m) & 15;
            pa0 = __builtin_amdgcn_exp2f(sacc[ss][kb2][r2]);
            pa1 = __builtin_amdgcn_exp2f(sacc[ss][kb2][r2 + 1]);
            if (m >= 1) pair_sum_pack(m - 1, q0_, q1_);
            const int pk = m >= 1 ? m - 1 : 0, kbk = pk >> 3, r = (2 * pk) & 15;
            asm volatile("" : "+v"(pw[ss][kbk][r >> 3]), "+v"(ps0), "+v"(ps1), "+v"(ps2), "+v"(ps3), "+v"(pa0), "+v"(pa1));
        };
        if (tail) {
            int lim = S - kv0 - 8 * hi;
            asm volatile("" : "+v"(lim));
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbk * 32 + 16 * (r >> 3) + (r & 7) >= lim) sacc[ss][kbk][r] = -1e30f;
        }
        if (need_max) {                      // classic online softmax (the redo pass): exact row maximum, subtracted before the pipelined region
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last S^T MFMA of the previous phase must have written its scores
            float tmax = sacc[ss][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sacc[ss][0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[ss][1][r]);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            const float tmc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            if (__any(tmc > mc[bs] + thr)) {
                const float mn = fmaxf(mc[bs], tmc);
                const float alpha = __builtin_amdgcn_exp2f(mc[bs] - mn);
                mc[bs] = mn;
                lrun[bs] *= alpha;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[bs][i][r] *= alpha;
            }
            const float mref = mc[bs];
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[ss][kbk][r] -= mref;
        }
        ps0 = ps1 = ps2 = ps3 = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            mfma_step(m);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ESME_W4_ABL & 1)) softmax_slot(m);
            if constexpr (!(ESME_W4_ABL & 2)) hook(m);
            __builtin_amdgcn_sched_barrier(0);
        }
        pair_sum_pack(15, pa0, pa1);
        const float psum = (ps0 + ps1) + (ps2 + ps3);
        if (__any(!(psum < 1e30f))) ovf = 1;
        lrun[bs] += psum;
    };
    using I0 = integral_constant<int, 0>; using I1 = integral_constant<int, 1>; using I2 = integral_constant<int, 2>; using I3 = integral_constant<int, 3>;
    using YES = integral_constant<bool, true>; using NO = integral_constant<bool, false>;

    const int nt = (S + KT - 1) / KT;
    bool exact = !a.spec;
    for (;;) {
#pragma unroll
        for (int bb = 0; bb < QB; ++bb) {
            mc[bb] = -1e30f; lrun[bb] = wave_active ? 0.f : 1.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[bb][i][r] = 0.f;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[st][i][r] = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) pw[st][i][s_] = u32x4{0u, 0u, 0u, 0u};
            }
        dma_k(0, smem);
        dma_v(0, smem);
        if (nt > 1) { dma_k(1, smem + SLOT); dma_v(1, smem + SLOT); }
        if (nt > 2) dma_k(2, smem + 2 * SLOT);
        {
            const u32x4 z = {0u, 0u, 0u, 0u};
            char* v3 = smem + 3 * SLOT + K_BYTES;
#pragma unroll
            for (int i = 0; i < (D * 128) / (NT * 16); ++i) *reinterpret_cast<u32x4*>(v3 + (i * NT + tid) * 16) = z;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the Q fragments re-defined while no load is in flight: hipcc's wait-count pass otherwise guards their first uses in
        // the loop with vmcnt(3..0), which drains the tile prefetch issued a few instructions earlier -- see attn_pp64_kernel)
#pragma unroll
        for (int bb = 0; bb < QB; ++bb)
            asm volatile("" : "+a"(qf[bb][0]), "+a"(qf[bb][1]), "+a"(qf[bb][2]), "+a"(qf[bb][3]) : : "memory");
        __syncthreads();
        if (wave_active) {
            // fragment sets: K(0); V^T of slot 3 (zeros: phase (0, 0) multiplies it by P = 0); S^T(b0, tile 0)
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) kfr[kbk][ds] = kfrag(smem, kbk, ds);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) vfr[db][ks] = vfrag(smem + 3 * SLOT, db, ks);
#pragma unroll
            for (int ds = 0; ds < DS; ++ds)
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk) {
                    if (ds == 0) ESME_MFMA32_VA0(sacc[0][kbk], kfr[kbk][ds], qf[0][ds]);
                    else ESME_MFMA32_VA(sacc[0][kbk], kfr[kbk][ds], qf[0][ds]);
                }
        }
        const bool ragged = (S & (KT - 1)) != 0;
        for (int t = 0; t < nt; ++t) {
            const bool pf_k = t + 3 < nt, pf_v = t + 2 < nt;
            char* kslot = smem + ((t + 3) & 3) * SLOT;
            char* vslot = smem + ((t + 2) & 3) * SLOT;
            if (wave_active) {
                const char* cur = smem + (t & 3) * SLOT;
                const char* nxt = smem + ((t + 1) & 3) * SLOT;
                const bool tail = t == nt - 1 && ragged;
                const bool need_max = exact;
                // phase (0, t): V^T set V(t-1) -> V(t) in place
                phase(I0{}, need_max, tail, t * KT, NO{}, nxt, YES{}, cur, [&](const int m) { if (m == 5 && pf_k) dma_piece(0, t + 3, kslot, 0); });
                phase(I1{}, need_max, tail, t * KT, NO{}, nxt, NO{}, cur, [&](const int m) { if (m == 5 && pf_k) dma_piece(0, t + 3, kslot, 1); });
                // phase (2, t): K set K(t) -> K(t+1) in place
                phase(I2{}, need_max, tail, t * KT, YES{}, nxt, NO{}, cur, [&](const int m) { if (m == 5 && pf_v) dma_piece(1, t + 2, vslot, 0); });
                phase(I3{}, need_max, tail, t * KT, NO{}, nxt, NO{}, cur, [&](const int m) { if (m == 5 && pf_v) dma_piece(1, t + 2, vslot, 1); });
            } else {
                if (pf_k) dma_k(t + 3, kslot);
                if (pf_v) dma_v(t + 2, vslot);
            }
            if (t + 3 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * KI) : "memory");
            else if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(KI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (wave_active) {                          // drain: O^T(b3) += V(nt-1) P(b3, nt-1)  (the V^T set still holds V(nt-1))
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int db = 0; db < 2; ++db) ESME_MFMA32_AV(oacc[3][db], vfr[db][ks], pw[1][ks >> 1][ks & 1]);
        }
        if (!exact) {
#pragma unroll
            for (int bb = 0; bb < QB; ++bb)
                if (__any(!(lrun[bb] > 1e-30f))) ovf = 1;
        }
        if (exact || !__syncthreads_or(ovf)) break;
        exact = true;
        ovf = 0;
    }
    if (!wave_active) return;

    // ---- epilogue: normalise, transpose through a wave-private LDS slab (4 KB per wave in a slot no wave reads any more), whole rows out
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // the drain MFMAs have written O^T
    char* slab = smem + ((nt + 1) & 3) * SLOT + wave * 4096;
#pragma unroll
    for (int bb = 0; bb < QB; ++bb) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lrun[bb]), __float_as_uint(lrun[bb]), false, false);
        const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
        if (bb) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk = {pack_bf16(oacc[bb][db][4 * g] * inv, oacc[bb][db][4 * g + 1] * inv),
                            pack_bf16(oacc[bb][db][4 * g + 2] * inv, oacc[bb][db][4 * g + 3] * inv)};
                *reinterpret_cast<u32x2*>(slab + l31 * 128 + (((db * 4 + g) ^ (l31 & 7)) << 4) + hi * 8) = pk;
            }
        __builtin_amdgcn_wave_barrier();
        const int rbase = wrow0 + bb * 32;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 8 + (lane >> 3), ch = lane & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(slab + r * 128 + ((ch ^ (r & 7)) << 4));
            if (rbase + r < S) *reinterpret_cast<u32x4*>(a.o + (int64_t)(s0 + rbase + r) * a.ldo + h * D + ch * 8) = v;
        }
    }
}

#endif  // ESME_ATTN_W4

}  // namespace esme

using namespace esme;

template <int NW, bool QP = false, int D = 64, bool F16 = false>
static int launch_pp64(AttnArgs& a, int B, int max_len, hipStream_t s) {
    constexpr int smem = 4 * (KT * D * 2 + D * 128);
    auto kern = attn_pp64_kernel<NW, QP, D, F16>;
    static std::atomic<unsigned long long> done{0ull};         // dynamic-LDS attribute: per (kernel, device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return fail(ESME_ERR_LAUNCH, "attn: cannot raise the dynamic LDS limit");
        done.fetch_or(bit, std::memory_order_release);
    }
    a.nqt = (max_len + NW * 64 - 1) / (NW * 64);
    const int64_t blocks = (int64_t)a.nqt * (((int64_t)a.H * B + 7) / 8) * 8;
    if (blocks > 0x7fffffffLL) return fail(ESME_ERR_UNSUPPORTED, "attn: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned int)blocks), dim3(NW * 64), smem, s, a);
    return check_launch("attn_varlen_fwd");
}

template <int D, bool F16, bool QKP, bool QP>
static int launch_sb(AttnSplitArgs& sa, int B, int max_len, hipStream_t s) {
    constexpr int smem = 3 * ((QKP ? 2 : 1) * KT * D * 2 + D * 128);
    auto kern = attn_sb_kernel<D, F16, QKP, QP>;
    static std::atomic<unsigned long long> done{0ull};         // dynamic-LDS attribute: per (kernel, device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return fail(ESME_ERR_LAUNCH, "attn: cannot raise the dynamic LDS limit");
        done.fetch_or(bit, std::memory_order_release);
    }
    sa.a.nqt = (max_len + 127) / 128;
    const int64_t blocks = (int64_t)sa.a.nqt * (((int64_t)sa.a.H * B + 7) / 8) * 8;
    if (blocks > 0x7fffffffLL) return fail(ESME_ERR_UNSUPPORTED, "attn: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned int)blocks), dim3(256), smem, s, sa);
    return check_launch("attn_varlen_fwd");
}

#ifdef ESME_ATTN_W4
static int launch_w4(AttnArgs& a, int B, int max_len, hipStream_t s) {
    constexpr int smem = 4 * (KT * 64 * 2 + 64 * 128);
    auto kern = attn_w4_kernel;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return fail(ESME_ERR_LAUNCH, "attn: cannot raise the dynamic LDS limit");
        done.fetch_or(bit, std::memory_order_release);
    }
    a.nqt = (max_len + 511) / 512;
    const int64_t blocks = (int64_t)a.nqt * (((int64_t)a.H * B + 7) / 8) * 8;
    if (blocks > 0x7fffffffLL) return fail(ESME_ERR_UNSUPPORTED, "attn: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned int)blocks), dim3(256), smem, s, a);
    return check_launch("attn_varlen_fwd");
}

#endif

// (per-call options, esme_attn_opts_t: no process-global tuning state; NULL = the defaults below)
static int attn_fwd(const void* q, const void* k, const void* v, int64_t ld_qkv, void* o, int64_t ld_o, const int32_t* cu_lens,
                    int B, int64_t T, int H, int d, int max_len, float softmax_scale, void* stream, bool exact,
                    const esme_attn_opts_t* opts) {
    ESME_CHECK_ARG(!opts || opts->struct_bytes == (int)sizeof(esme_attn_opts_t), "attn: options struct of another ABI");
    const int g_attn_variant = opts ? opts->variant : 0;       // 0 = heuristic, 1 = first-generation kernel, 4 / 8 = ping-pong with 4 / 8 waves
    const int g_force_qb = opts ? opts->q_blocks : 0;          // first-generation kernel: q-blocks per wave (0 = heuristic)
    const float g_attn_thr = opts ? opts->defer_max_thr : 8.0f;   // defer-max threshold, log2 units
    const int g_attn_spec = opts ? opts->speculative : 1;      // speculative softmax in the ping-pong kernel
    ESME_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && d > 0 && max_len >= 0, "attn: bad sizes");
    if (T == 0 || B == 0) return ESME_OK;
    ESME_CHECK_ARG(q && k && v && o && cu_lens, "attn: null pointer");
    ESME_CHECK_ARG(ld_qkv % 8 == 0 && ld_qkv >= (int64_t)H * d && ld_o % 4 == 0 && ld_o >= (int64_t)H * d,
                   "attn: bad row strides");
    ESME_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && (reinterpret_cast<uintptr_t>(o) & 7u) == 0,
                   "attn: misaligned");
    ESME_CHECK_ARG(max_len > 0 && H <= 65535 && B <= 65535, "attn: max_len must be > 0, H and B <= 65535");
    // q_prescaled: q already carries softmax_scale * log2(e) (esme_gemm_fusion_t.q_scale): every kernel then runs with c = 1, and
    // the 4-wave head-dim-64 kernel in its no-reference-maximum form
    const bool f16 = opts && opts->f16;                         // fp16 operands: speculative / defer-max passes bounded to fp16's range (see the kernels)
    const bool qp = opts && opts->q_prescaled;
    ESME_CHECK_ARG(!(f16 && qp) || ((d == 64 || d == 32) && g_attn_variant != 1 && g_attn_variant != 2 && ld_o % 8 == 0 && aligned16(o)),
                   "attn: fp16 operands combine with q_prescaled only in the ping-pong kernel (head dims 64 / 32: fixed reference 4, redo outside fp16's range)");
    AttnArgs a{(const u16*)q, (const u16*)k, (const u16*)v, ld_qkv, (u16*)o, ld_o, cu_lens, H,
               qp ? 1.0f : softmax_scale * 1.4426950408889634f, 1, H * B, exact ? 0.0f : g_attn_thr, exact ? 0 : g_attn_spec,
               opts ? opts->seq_order : nullptr};
    const hipStream_t s = (hipStream_t)stream;
    // (the ping-pong kernel addresses K / V with 32-bit byte offsets inside one sequence: (max_len + one tile) rows must fit)
    const bool fits32 = ((int64_t)max_len + KT) * ld_qkv * 2 < 0xffffffffLL;
    if ((d == 64 || d == 32) && g_attn_variant == 2 && ld_o % 8 == 0 && aligned16(o) && fits32) {
        // per-call option 2: the single-block pipelined kernel (attn_sb_kernel: 128 query rows per workgroup) -- what the q/k-pair entry runs; here for
        // A/B measurements of the plain forms against the ping-pong kernel
        AttnSplitArgs sa{a, 0, 0};
        if (d == 64) return f16 ? launch_sb<64, true, false, false>(sa, B, max_len, s) : (qp ? launch_sb<64, false, false, true>(sa, B, max_len, s) : launch_sb<64, false, false, false>(sa, B, max_len, s));
        return f16 ? launch_sb<32, true, false, false>(sa, B, max_len, s) : (qp ? launch_sb<32, false, false, true>(sa, B, max_len, s) : launch_sb<32, false, false, false>(sa, B, max_len, s));
    }
    if (d == 64 && g_attn_variant != 1 && ld_o % 8 == 0 && aligned16(o) && fits32) {
        // head dim 64 (ESM2-650M / 3B, ESM-C): the software-pipelined kernel.  4 waves = 256 query rows per workgroup, two
        // workgroups per CU (one's prologue / epilogue overlaps the other's main loop): measured faster than 8 waves
        // (one workgroup per CU) from S = 130 to S = 2 000; the 8-wave form stays behind the tuning hook.
        const int nw = g_attn_variant == 8 ? 8 : 4;
        if (f16) return qp ? launch_pp64<4, true, 64, true>(a, B, max_len, s) : launch_pp64<4, false, 64, true>(a, B, max_len, s);
#ifdef ESME_ATTN_W4
        if (qp && g_attn_variant == 16) return launch_w4(a, B, max_len, s);          // one wave per SIMD, four q-blocks per wave (lab build)
#endif
        if (qp && nw == 4) return launch_pp64<4, true>(a, B, max_len, s);
        return nw == 8 ? launch_pp64<8>(a, B, max_len, s) : launch_pp64<4>(a, B, max_len, s);
    }
    if (d == 32 && g_attn_variant != 1 && ld_o % 8 == 0 && aligned16(o) && fits32) {
        // head dim 32 (ESM2-150M; ESM2-35M's padded heads): the same software-pipelined kernel at D = 32 (round 4)
        if (f16) return qp ? launch_pp64<4, true, 32, true>(a, B, max_len, s) : launch_pp64<4, false, 32, true>(a, B, max_len, s);
        return qp ? launch_pp64<4, true, 32>(a, B, max_len, s) : launch_pp64<4, false, 32>(a, B, max_len, s);
    }
    // two 32-row q-blocks per wave when the longest sequence fills at least one 256-row tile
    // (head dim 128 keeps one: its accumulators alone take 128 VGPRs per q-block)
    const int qb = (g_force_qb ? g_force_qb : (max_len >= 192 ? 2 : 1));
    const bool two = qb == 2 && d <= 64;
    const int rows = QT * (two ? 2 : 1);
    const dim3 grid((unsigned int)((max_len + rows - 1) / rows), (unsigned int)H, (unsigned int)B), block(256);
#define ESME_ATTN(DD)                                                                         \
    case DD:                                                                                   \
        if (f16) {                                                                             \
            if (two) hipLaunchKernelGGL((attn_varlen_kernel<DD, (DD <= 64 ? 2 : 1), true>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((attn_varlen_kernel<DD, 1, true>), grid, block, 0, s, a);  \
        } else if (two) hipLaunchKernelGGL((attn_varlen_kernel<DD, (DD <= 64 ? 2 : 1)>), grid, block, 0, s, a); \
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

template <int D, bool F16 = false, bool QKP = false>
static int launch_split(const AttnSplitArgs& sa, const dim3 grid, hipStream_t s) {
    constexpr int smem = 2 * (2 * KT * D * 2 + (QKP ? 1 : 2) * D * 128);
    auto kern = attn_split_kernel<D, F16, QKP>;
    if (smem >= 64 * 1024) {
        static std::atomic<unsigned long long> done{0ull};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
                return fail(ESME_ERR_LAUNCH, "attn_split: cannot raise the dynamic LDS limit");
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, sa);
    return check_launch("attn_varlen_fwd_split");
}

extern "C" int esme_hip_attn_varlen_fwd_split(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t lo_qkv, void* o,
                                              int64_t ld_o, int64_t lo_o, const int32_t* cu_lens, int B, int64_t T, int H, int d,
                                              int max_len, float softmax_scale, const int32_t* seq_order, void* stream) {
    ESME_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && d > 0 && max_len >= 0, "attn_split: bad sizes");
    if (T == 0 || B == 0) return ESME_OK;
    ESME_CHECK_ARG(q && k && v && o && cu_lens, "attn_split: null pointer");
    ESME_CHECK_ARG(ld_qkv % 8 == 0 && lo_qkv % 8 == 0 && lo_qkv > 0 && ld_o % 4 == 0 && lo_o % 4 == 0 && lo_o >= (int64_t)H * d &&
                   ld_o >= lo_o + (int64_t)H * d, "attn_split: bad row strides / pair offsets");
    ESME_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && (reinterpret_cast<uintptr_t>(o) & 7u) == 0, "attn_split: misaligned");
    ESME_CHECK_ARG(max_len > 0 && H <= 65535 && B <= 65535, "attn_split: max_len must be > 0, H and B <= 65535");
    AttnSplitArgs sa{{(const u16*)q, (const u16*)k, (const u16*)v, ld_qkv, (u16*)o, ld_o, cu_lens, H, softmax_scale * 1.4426950408889634f, 1, H * B,
                      0.0f, 0, seq_order}, lo_qkv, lo_o};
    const dim3 grid((unsigned int)((max_len + QT - 1) / QT), (unsigned int)H, (unsigned int)B);
    const hipStream_t s = (hipStream_t)stream;
    switch (d) {
        case 16: return launch_split<16>(sa, grid, s);
        case 32: return launch_split<32>(sa, grid, s);
        case 64: return launch_split<64>(sa, grid, s);
        case 128: return launch_split<128>(sa, grid, s);
        default: ESME_FAIL(ESME_ERR_UNSUPPORTED, "attn_split: head dim must be 16, 32, 64 or 128");
    }
}

extern "C" int esme_hip_attn_varlen_fwd_qkpair_f16_opts(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t lo_qk, void* o,
                                                        int64_t ld_o, const int32_t* cu_lens, int B, int64_t T, int H, int d,
                                                        int max_len, float softmax_scale, const esme_attn_opts_t* opts, void* stream) {
    ESME_CHECK_ARG(!opts || opts->struct_bytes == (int)sizeof(esme_attn_opts_t), "attn_qkpair: options struct of another ABI");
    const int32_t* seq_order = opts ? opts->seq_order : nullptr;
    ESME_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && d > 0 && max_len >= 0, "attn_qkpair: bad sizes");
    if (T == 0 || B == 0) return ESME_OK;
    ESME_CHECK_ARG(q && k && v && o && cu_lens, "attn_qkpair: null pointer");
    ESME_CHECK_ARG(ld_qkv % 8 == 0 && lo_qk % 8 == 0 && lo_qk > 0 && ld_o % 4 == 0 && ld_o >= (int64_t)H * d, "attn_qkpair: bad row strides / pair offset");
    ESME_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && (reinterpret_cast<uintptr_t>(o) & 7u) == 0, "attn_qkpair: misaligned");
    ESME_CHECK_ARG(max_len > 0 && H <= 65535 && B <= 65535, "attn_qkpair: max_len must be > 0, H and B <= 65535");
    AttnSplitArgs sa{{(const u16*)q, (const u16*)k, (const u16*)v, ld_qkv, (u16*)o, ld_o, cu_lens, H, softmax_scale * 1.4426950408889634f, 1, H * B,
                      0.0f, 0, seq_order}, lo_qk, 0};
    const dim3 grid((unsigned int)((max_len + QT - 1) / QT), (unsigned int)H, (unsigned int)B);
    const hipStream_t s = (hipStream_t)stream;
    // options variant 2, head dims 64 / 32: the key-axis-pipelined kernel (round 6; exact maxima: spec = 0, thr = 0).  Measured (profiles/r06_attn_sb_bench.txt):
    // 404 vs 363 us at 100 x 500, 636 vs 641 at 49 x 1 002, 1 140 vs 1 168 at 25 x 2 000 -- not the hoped-for 260 us, so the first-generation kernel stays the default
    const bool fits32 = ((int64_t)max_len + KT) * ld_qkv * 2 < 0xffffffffLL;
    if ((d == 64 || d == 32) && opts && opts->variant == 2 && ld_o % 8 == 0 && aligned16(o) && fits32)
        return d == 64 ? launch_sb<64, true, true, false>(sa, B, max_len, s) : launch_sb<32, true, true, false>(sa, B, max_len, s);
    switch (d) {
        case 16: return launch_split<16, true, true>(sa, grid, s);
        case 32: return launch_split<32, true, true>(sa, grid, s);
        case 64: return launch_split<64, true, true>(sa, grid, s);
        default: ESME_FAIL(ESME_ERR_UNSUPPORTED, "attn_qkpair: head dim must be 16, 32 or 64");
    }
}

extern "C" int esme_hip_attn_varlen_fwd_qkpair_f16(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t lo_qk, void* o,
                                                   int64_t ld_o, const int32_t* cu_lens, int B, int64_t T, int H, int d,
                                                   int max_len, float softmax_scale, const int32_t* seq_order, void* stream) {
    esme_attn_opts_t o1{(int)sizeof(esme_attn_opts_t), 0, 0, 0.0f, 0, seq_order, 0, 1};
    return esme_hip_attn_varlen_fwd_qkpair_f16_opts(q, k, v, ld_qkv, lo_qk, o, ld_o, cu_lens, B, T, H, d, max_len, softmax_scale, &o1, stream);
}

extern "C" int esme_hip_attn_varlen_fwd(const void* q, const void* k, const void* v, int64_t ld_qkv, void* o,
                                        int64_t ld_o, const int32_t* cu_lens, int B, int64_t T, int H, int d,
                                        int max_len, float softmax_scale, void* stream) {
    return attn_fwd(q, k, v, ld_qkv, o, ld_o, cu_lens, B, T, H, d, max_len, softmax_scale, stream, false, nullptr);
}

extern "C" int esme_hip_attn_varlen_fwd_opts(const void* q, const void* k, const void* v, int64_t ld_qkv, void* o,
                                             int64_t ld_o, const int32_t* cu_lens, int B, int64_t T, int H, int d,
                                             int max_len, float softmax_scale, const esme_attn_opts_t* opts, void* stream) {
    return attn_fwd(q, k, v, ld_qkv, o, ld_o, cu_lens, B, T, H, d, max_len, softmax_scale, stream, false, opts);
}

extern "C" int esme_hip_attn_varlen_fwd_exact(const void* q, const void* k, const void* v, int64_t ld_qkv, void* o,
                                              int64_t ld_o, const int32_t* cu_lens, int B, int64_t T, int H, int d,
                                              int max_len, float softmax_scale, void* stream) {
    return attn_fwd(q, k, v, ld_qkv, o, ld_o, cu_lens, B, T, H, d, max_len, softmax_scale, stream, true, nullptr);
}
