// 256 x 256 persistent implicit-GEMM kernel for the wide bf16 convolutions (gfx950).
//
//   C[m][n] = scale * sum_{tap,k} A[m + tap - pad][k] * W[tap][n][k] + bias[n] (+ addend[m][n])
//
// Same contraction as gemm_nt (gemm.hip); reference call sites: modules/decoder.py:117-121 (recon head 1024 -> 95008),
// modules/common.py:135-141 (decoder residual block 1x1 / k5 convs), modules/encoder.py:34,119-121 (95008 -> 1024) and
// their input gradients.
//
// Structure (one 512-thread workgroup per CU, two waves per SIMD):
//  * tile 256 (m) x 256 (n), K-tile 64; wave (g, w) = (row half, 64-column block) owns 128 x 64 outputs: 8 x 4 tiles of
//    v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the MFMA's first operand, so a lane ends up with four consecutive
//    n of one row m (a 16-byte bf16 chunk after one v_permlane16_swap, stored straight from registers: no LDS epilogue).
//  * operands go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 128-byte rows, XOR swizzle on the source side
//    (slot p of row r holds source chunk p ^ ((r >> 1) & 7): conflict-free ds_read_b128 for the 16x16x32 fragments);
//    masked taps / K tails / tile edges are zero-filled by the buffer range check.
//  * LDS = two K-tile buffers of 64 KiB, refilled in QUARTERS: a K-tile is consumed in four phases
//      L1: read A rows 0-63 + W cols 0-31   M1: 16 MFMA        L2: read W cols 32-63      M2: 16 MFMA
//      L3: read A rows 64-127               M3: 16 MFMA        L4: (no reads)             M4: 16 MFMA
//    and the quarter a phase has read is refilled two phases later with the data of K-tile t+1 / t+2, so every DMA has a
//    whole K-tile of MFMA time to land (`s_waitcnt vmcnt(8)` + s_barrier at the end of each L section: only the DMAs of
//    the last four sections may still be in flight; the barrier in between orders every wave's reads before the refill).
//  * per-lane DMA offsets are loop constants; the moving part (tap * lda + k) is the instruction's scalar offset, and the
//    steady state runs a conditional-free body unrolled over the two buffers (compile-time LDS addresses).
//  * persistent: a workgroup walks its list of (tile, split-K slice) work items with the DMA stream running two K-tiles
//    ahead ACROSS item boundaries, so prologue latency is paid once per workgroup, and the epilogue of one item runs
//    while the first K-tiles of the next are landing.
//  * deterministic: split-K slices go to fp32 slabs combined in a fixed order; GroupNorm statistics are per-(item, wave)
//    partial sums reduced by t256_stats_finalize in a fixed order (no atomics).
#include <math.h>
#include <stdlib.h>
#include "sgv_common.h"

typedef __attribute__((address_space(3))) void t256_lds_t;
typedef float t256_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t t256_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t t256_u2 __attribute__((ext_vector_type(2)));
constexpr uint32_t T256_OOB = 0x7FFFFFF0u;

#ifndef T256_STAGGER
#define T256_STAGGER 1       // 1: the two row halves of a workgroup run half a section out of phase (see T256_KTILE)
#endif
#ifdef T256_ABL_NOBAR
#define T256_ABL_NOBAR_ 1
#else
#define T256_ABL_NOBAR_ 0
#endif
#ifndef T256_FINEWAIT
#define T256_FINEWAIT 1
#endif
#ifndef T256_DELAY
#define T256_DELAY 0         // start skew: workgroup slot j of an XCD sleeps j * T256_DELAY * 64 cycles (de-phases the epilogue write bursts)
#endif
#ifdef T256_ABL_NOSTORE
#define T256_ABL_NOSTORE_ 1
#else
#define T256_ABL_NOSTORE_ 0
#endif
#ifndef T256_SETPRIO
#define T256_SETPRIO 1
#endif
#ifndef T256_STRM_DEFAULT
#define T256_STRM_DEFAULT 0   // stream policy of the banded launches (see STRM): measured, not assumed
#endif

// one 16-byte bf16 chunk from two packed halves
__device__ __forceinline__ uint32_t t256_pack2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float t256_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float t256_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }

// OUT: 0 = bf16 output (scale, bias, addend, optional statistics), 1 = fp32 (split-K slab: raw sums; or final fp32 output)
// MT:  the layer has more than one tap (per-row tap windows are tested when a DMA is issued)
// C2D: 2-D taps (GemmNT::cv_*, needs MT): rows are output pixels of a channels-last image batch, tap = (kh, kw)
// STRM: cache policy of the two streams of a launch whose activation tiles should stay in the XCD's L2 (the recon head: every
//       weight panel is read once per XCD and 608 MB of output flow through the same 4 MiB): bit 0 = the weight stream is
//       loaded non-temporally, bit 1 = the bf16 output is stored with sc1 (the line leaves the L2 once written)
// TS:   tile shape.  0 = 256 (m) x 256 (n): wave (g, wq) = (row half, 64-column block).  1 = 128 (m) x 512 (n): all eight waves share
//       the 128 rows and own 64 columns each -- M = 3200 is 25 such row tiles, so 25 x N/512 x slices fill ONE round of the chip with
//       no 128-row tail launch.  Same sections and refill order; a K-tile is 16 KiB of activations + 64 KiB of weights (80 KiB per
//       buffer, both buffers = the CU's whole LDS), the quarters are A0 / A1 = rows 0-63 / 64-127 (one DMA piece per wave) and
//       B0 / B1 = columns 0-31 / 32-63 of every wave's block (four pieces per wave, its OWN weight rows), so any four consecutive
//       sections issue 1 + 1 + 4 + 4 = 10 pieces per wave: the counted wait is vmcnt(10) where the square tile has vmcnt(8).
template <int OUT, bool MT, bool C2D = false, int STRM = 0, int TS = 0>
__global__ __launch_bounds__(512) void gemm_nt_t256_kernel(const GemmNT p) {
    constexpr int ESZ = 2;
    constexpr int NST = (OUT == 0 ? 16 : 32) + 1;     // buffer stores per wave and epilogue + the next item's bias load
    constexpr int BUFB = TS ? 81920 : 65536;          // bytes of one K-tile buffer
    constexpr int WOFF = TS ? 16384 : 32768;          // the weight region's offset inside a buffer
    constexpr int VMW = TS ? 10 : 8;                  // DMA pieces a wave issues in four consecutive sections
    constexpr int TMS = TS ? 7 : 8, TNS = TS ? 9 : 8; // log2 of the tile's rows / columns
    static_assert(!(TS && C2D), "the 128 x 512 tile has no 2-D tap mode");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUFB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wq = wave & 3;

    // ---------------- static schedule: XCD-chunked list of work items ----------------
    const int tiles_m = (p.M + (1 << TMS) - 1) >> TMS, tiles_n = (p.N + (1 << TNS) - 1) >> TNS;
    const int ntile = tiles_m * tiles_n;
    const int kchunks = (p.K + 63) >> 6;
    const int total_kt = p.taps * kchunks;
    const int nitems = ntile * p.splitk;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int q8 = nitems >> 3, r8 = nitems & 7;
    const int it_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int it_hi = it_lo + (xcd < r8 ? q8 + 1 : q8);
    const int nbx = ((int)gridDim.x - xcd + 7) >> 3;          // workgroups carrying this XCD label
    if (it_lo + jb >= it_hi) return;                          // workgroup-uniform
    if (T256_DELAY > 0 && q8 >= nbx) for (int d_ = 0; d_ < jb; ++d_) __builtin_amdgcn_s_sleep(T256_DELAY);

    const int lda_b = (int)(p.lda * ESZ), ldw_b = (int)(p.ldw * ESZ), wts_b = (int)(p.w_tap_stride * ESZ);
    const int kK_b = p.K * ESZ;
    // the A descriptor starts `pad` rows BEFORE the buffer: per-lane offsets (row * lda) stay non-negative and the uniform part
    // (tap * lda + k) goes to the instruction's scalar offset.  Lanes whose tap leaves the sample window get an offset beyond
    // the extent (hardware zero-fill), so nothing in front of the buffer is ever read.
    // (C2D: the shift is cv_P image rows + cv_P pixels, a lane's row is the pixel (b, oh*S, ow*S), the uniform part (kh, kw).)
    const long a_shift = C2D ? ((long)p.cv_P * p.cv_W + p.cv_P) * lda_b : (long)p.pad * lda_b;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.A)) - a_shift, 0, (int)(p.a_bytes + a_shift), 0x00020000);
    const int wstep_b = (C2D && p.cv_flip) ? -wts_b : wts_b;            // cv_flip: tap j multiplies W[taps - 1 - j]
    const int w0_b = (C2D && p.cv_flip) ? (p.taps - 1) * wts_b : 0;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (int)p.w_bytes, 0x00020000);
    const float sc = p.scale ? *p.scale : 1.0f;

    // ---------------- DMA roles ----------------
    // one op = 8 rows x 128 B.  Wave (g, wq): A rows g*128 + {0, 64} + wq*16 + {0, 8} + (lane >> 3) of the tile (its own row half),
    // W rows wsel*64 + {0, 32} + (wq & 1)*16 + {0, 8} + (lane >> 3) with wsel = 2g + (wq >> 1).
    // TS = 1: one op = 8 rows x 128 B as well.  Wave w: A rows {0, 64} + w*8 + (lane >> 3) (one op per quarter), W rows w*64 +
    // {0, 32} + {0, 8, 16, 24} + (lane >> 3) (four ops per quarter: its own 64 weight rows).
    const int rl = lane >> 3, dp = lane & 7;
    const int a_row0 = TS ? wave * 8 : g * 128 + wq * 16;      // + 64 for the A1 quarter, (TS = 0) + 8 for the second op
    const int w_row0 = TS ? wave * 64 : (2 * g + (wq >> 1)) * 64 + (wq & 1) * 16;
    // source chunk of an op: dp ^ ((row >> 1) & 7) with row = base (multiple of 16) + {0, 8} + rl
    const int dc0 = dp ^ ((rl >> 1) & 7), dc1 = dp ^ (((8 + rl) >> 1) & 7);
    const int dcA = TS ? dp ^ ((((wave & 1) << 2) + (rl >> 1)) & 7) : dc0;       // TS = 1: the A op's rows start at a multiple of 8
    unsigned char* const ldsA0 = smem + a_row0 * 128;
    unsigned char* const ldsA1 = smem + (a_row0 + 64) * 128;
    unsigned char* const ldsB0 = smem + WOFF + w_row0 * 128;
    unsigned char* const ldsB1 = smem + WOFF + (w_row0 + 32) * 128;

    // per-item per-lane DMA state
    uint32_t vA0, vA1, vA2, vA3;          // A0 op 0/1, A1 op 0/1: (m0 + row) * lda_b + dc * 16 (relative to the shifted base)
    uint32_t imA0, imA1, imA2, imA3;      // bit j set: tap j of that row leaves the sample window
    uint32_t vW0, vW1, vW2, vW3;          // B0 op 0/1, B1 op 0/1: (n0 + row) * ldw_b + dc * 16, or out of range
    uint32_t vW4 = 0, vW5 = 0, vW6 = 0, vW7 = 0;     // TS = 1: B0 ops 0-3 = vW0, vW1, vW4, vW5; B1 ops 0-3 = vW2, vW3, vW6, vW7
    // load cursor (wave-uniform)
    int li = it_lo + jb;                   // item being loaded
    int l_kt = 0, l_kt_end = 0;            // K-tile of item li the cursor points at / end of its slice
    int ld_j = 0, ld_kcb = 0;              // its tap and channel-chunk byte offset
    int ld_jh = 0, ld_jw = 0;              // C2D: (kh, kw) of tap ld_j
    int sA = 0, sW = 0;                    // scalar offsets: tap * lda + k / tap * tap_stride + k (bytes)
    bool l_active = true;

#define T256_UNI(X) __builtin_amdgcn_readfirstlane(X)
#define T256_ROWMASK(DST, AM)                                                                                 \
    {                                                                                                         \
        uint32_t mk_ = 0u;                                                                                    \
        const int at_ = ((AM) + p.trow0) % p.Tlen;                                                            \
        for (int j_ = 0; j_ < p.taps; ++j_)                                                                   \
            if ((unsigned)(at_ + j_ - p.pad) >= (unsigned)p.Tlen) mk_ |= 1u << j_;                            \
        DST = mk_;                                                                                            \
    }
    // C2D: offset of the row's pixel (b, oh*S, ow*S) and the taps whose pixel leaves the image (all of them for rows >= M)
#define T256_ROW2D(V, IM, AM, DC)                                                                             \
    {                                                                                                         \
        const int hw_ = p.cv_Ho * p.cv_Wo;                                                                    \
        const int b_ = (AM) / hw_, q_ = (AM) - b_ * hw_;                                                      \
        const int oh_ = q_ / p.cv_Wo, ow_ = q_ - oh_ * p.cv_Wo;                                               \
        const int ih_ = oh_ * p.cv_S - p.cv_P, iw_ = ow_ * p.cv_S - p.cv_P;                                   \
        V = (uint32_t)((((long)b_ * p.cv_H + oh_ * p.cv_S) * p.cv_W + ow_ * p.cv_S) * lda_b + (DC) * 16);     \
        uint32_t mk_ = 0u;                                                                                    \
        for (int j_ = 0, kh_ = 0, kw_ = 0; j_ < p.taps; ++j_) {                                               \
            if ((AM) >= p.M || (unsigned)(ih_ + kh_) >= (unsigned)p.cv_H || (unsigned)(iw_ + kw_) >= (unsigned)p.cv_W) mk_ |= 1u << j_; \
            if (++kw_ == p.cv_kw) { kw_ = 0; ++kh_; }                                                         \
        }                                                                                                     \
        IM = mk_;                                                                                             \
    }
    // item -> (slice z, row tile tm, column tile tn).  p.band == 0: column-panel-major (the tiles_m row tiles of a weight panel are
    // neighbours).  p.band > 0: the row tiles are cut into bands of p.band; band-major, then column tile, row tile fastest -- an
    // XCD then works on one band (its activation tiles stay in L2) and streams the weight panels past it.
#define T256_TILE_OF(IT, Z, TM, TN)                                                                           \
    {                                                                                                         \
        Z = T256_UNI((IT) / ntile);                                                                           \
        const int rem_ = (IT) - Z * ntile;                                                                    \
        if (p.band > 0) {                                                                                     \
            const int bf_ = p.band * tiles_n;                                                                 \
            const int b_ = T256_UNI(rem_ / bf_), r1_ = rem_ - b_ * bf_;                                       \
            const int first_ = b_ * p.band, g_ = min(p.band, tiles_m - first_);                               \
            TN = T256_UNI(r1_ / g_); TM = first_ + (r1_ - TN * g_);                                           \
        } else {                                                                                              \
            TN = T256_UNI(rem_ / tiles_m); TM = rem_ - TN * tiles_m;                                          \
        }                                                                                                     \
    }
#define T256_SETUP_ITEM()                                                                                     \
    {                                                                                                         \
        int z_, tm_, tn_;                                                                                     \
        T256_TILE_OF(li, z_, tm_, tn_)                                                                        \
        const int m0_ = tm_ << TMS, n0_ = tn_ << TNS;                                                         \
        l_kt = T256_UNI((int)((long)total_kt * z_ / p.splitk));                                               \
        l_kt_end = T256_UNI((int)((long)total_kt * (z_ + 1) / p.splitk));                                     \
        const int kci_ = T256_UNI(l_kt / p.taps);                                                             \
        ld_j = l_kt - kci_ * p.taps;                                                                          \
        ld_kcb = kci_ * 128;                                                                                  \
        const int ar_ = m0_ + a_row0 + rl;                                                                    \
        if constexpr (C2D) {                                                                                  \
            ld_jh = T256_UNI(ld_j / p.cv_kw); ld_jw = ld_j - ld_jh * p.cv_kw;                                 \
            sA = (ld_jh * p.cv_W + ld_jw) * lda_b + ld_kcb; sW = w0_b + ld_j * wstep_b + ld_kcb;              \
            T256_ROW2D(vA0, imA0, ar_, dc0) T256_ROW2D(vA1, imA1, ar_ + 8, dc1)                               \
            T256_ROW2D(vA2, imA2, ar_ + 64, dc0) T256_ROW2D(vA3, imA3, ar_ + 72, dc1)                         \
        } else if constexpr (TS == 1) {                                                                       \
            sA = ld_j * lda_b + ld_kcb; sW = ld_j * wts_b + ld_kcb;                                           \
            vA0 = (uint32_t)((long)ar_ * lda_b + dcA * 16);                                                   \
            vA2 = (uint32_t)((long)(ar_ + 64) * lda_b + dcA * 16);                                            \
            vA1 = vA0; vA3 = vA2;                                                                             \
            if (MT) { T256_ROWMASK(imA0, ar_) T256_ROWMASK(imA2, ar_ + 64) imA1 = imA0; imA3 = imA2; }        \
        } else {                                                                                              \
            sA = ld_j * lda_b + ld_kcb; sW = ld_j * wts_b + ld_kcb;                                           \
            vA0 = (uint32_t)((long)ar_ * lda_b + dc0 * 16);                                                   \
            vA1 = (uint32_t)((long)(ar_ + 8) * lda_b + dc1 * 16);                                             \
            vA2 = (uint32_t)((long)(ar_ + 64) * lda_b + dc0 * 16);                                            \
            vA3 = (uint32_t)((long)(ar_ + 72) * lda_b + dc1 * 16);                                            \
            if (MT) { T256_ROWMASK(imA0, ar_) T256_ROWMASK(imA1, ar_ + 8) T256_ROWMASK(imA2, ar_ + 64) T256_ROWMASK(imA3, ar_ + 72) } \
        }                                                                                                     \
        const int wr_ = n0_ + w_row0 + rl;                                                                    \
        vW0 = wr_ < p.N ? (uint32_t)((long)wr_ * ldw_b + dc0 * 16) : 0x80000000u;                             \
        vW1 = wr_ + 8 < p.N ? (uint32_t)((long)(wr_ + 8) * ldw_b + dc1 * 16) : 0x80000000u;                   \
        vW2 = wr_ + 32 < p.N ? (uint32_t)((long)(wr_ + 32) * ldw_b + dc0 * 16) : 0x80000000u;                 \
        vW3 = wr_ + 40 < p.N ? (uint32_t)((long)(wr_ + 40) * ldw_b + dc1 * 16) : 0x80000000u;                 \
        if constexpr (TS == 1) {                                                                              \
            vW4 = wr_ + 16 < p.N ? (uint32_t)((long)(wr_ + 16) * ldw_b + dc0 * 16) : 0x80000000u;             \
            vW5 = wr_ + 24 < p.N ? (uint32_t)((long)(wr_ + 24) * ldw_b + dc1 * 16) : 0x80000000u;             \
            vW6 = wr_ + 48 < p.N ? (uint32_t)((long)(wr_ + 48) * ldw_b + dc0 * 16) : 0x80000000u;             \
            vW7 = wr_ + 56 < p.N ? (uint32_t)((long)(wr_ + 56) * ldw_b + dc1 * 16) : 0x80000000u;             \
        }                                                                                                     \
    }
    // per-lane offset of an A op: the row's offset, pushed out of range when the cursor's tap leaves the row's window
    // (v_bfe_i32 + v_and_or_b32); offsets + scalar offsets stay below 2^31, so an invalid lane stays >= 2^31 after the add
#define T256_VA(V, IM) (MT ? ((V) | ((uint32_t)__builtin_amdgcn_sbfe((int)(IM), (unsigned)ld_j, 1u) & 0x80000000u)) : (V))
    // K tail (last channel chunk of a K that is not a multiple of 64): chunks past K are zero-filled (general path only)
#define T256_KT(V, DC) ((ld_kcb + (DC) * 16 < kK_b) ? (V) : 0x80000000u)
#ifdef T256_ABL_NODMA
#define T256_DMA(RS, VOFF, SOFF, DST) asm volatile("; no dma %0 %1" :: "v"(VOFF), "s"(SOFF));
#else
#define T256_DMA(RS, VOFF, SOFF, DST) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (t256_lds_t*)(DST), 16, (VOFF), (SOFF), 0, 0);
#endif
#define T256_DMAW(VOFF, DST)                                                                                  \
    if constexpr ((STRM & 1) != 0) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (t256_lds_t*)(DST), 16, (VOFF), sW, 0, 2); } \
    else { T256_DMA(rsW, VOFF, sW, DST) }
    // quarter issues: F = 1 fast path (cursor active, no K tail), F = 0 general; PB = byte offset of the parity buffer filled
#define T256_ISSUE_A0(PB, F)                                                                                  \
    if ((F) || l_active) {                                                                                    \
        if constexpr (TS == 1) {                                                                              \
            if ((F) || ld_kcb + 128 <= kK_b) { T256_DMA(rsA, T256_VA(vA0, imA0), sA, ldsA0 + (PB)) }          \
            else { T256_DMA(rsA, T256_KT(T256_VA(vA0, imA0), dcA), sA, ldsA0 + (PB)) }                        \
        } else if ((F) || ld_kcb + 128 <= kK_b) { T256_DMA(rsA, T256_VA(vA0, imA0), sA, ldsA0 + (PB)) T256_DMA(rsA, T256_VA(vA1, imA1), sA, ldsA0 + (PB) + 1024) } \
        else { T256_DMA(rsA, T256_KT(T256_VA(vA0, imA0), dc0), sA, ldsA0 + (PB)) T256_DMA(rsA, T256_KT(T256_VA(vA1, imA1), dc1), sA, ldsA0 + (PB) + 1024) } \
    }
#define T256_ISSUE_A1(PB, F)                                                                                  \
    if ((F) || l_active) {                                                                                    \
        if constexpr (TS == 1) {                                                                              \
            if ((F) || ld_kcb + 128 <= kK_b) { T256_DMA(rsA, T256_VA(vA2, imA2), sA, ldsA1 + (PB)) }          \
            else { T256_DMA(rsA, T256_KT(T256_VA(vA2, imA2), dcA), sA, ldsA1 + (PB)) }                        \
        } else if ((F) || ld_kcb + 128 <= kK_b) { T256_DMA(rsA, T256_VA(vA2, imA2), sA, ldsA1 + (PB)) T256_DMA(rsA, T256_VA(vA3, imA3), sA, ldsA1 + (PB) + 1024) } \
        else { T256_DMA(rsA, T256_KT(T256_VA(vA2, imA2), dc0), sA, ldsA1 + (PB)) T256_DMA(rsA, T256_KT(T256_VA(vA3, imA3), dc1), sA, ldsA1 + (PB) + 1024) } \
    }
#define T256_ISSUE_B0(PB, F)                                                                                  \
    if ((F) || l_active) {                                                                                    \
        if ((F) || ld_kcb + 128 <= kK_b) {                                                                    \
            T256_DMAW(vW0, ldsB0 + (PB)) T256_DMAW(vW1, ldsB0 + (PB) + 1024)                                  \
            if constexpr (TS == 1) { T256_DMAW(vW4, ldsB0 + (PB) + 2048) T256_DMAW(vW5, ldsB0 + (PB) + 3072) } \
        } else {                                                                                              \
            T256_DMAW(T256_KT(vW0, dc0), ldsB0 + (PB)) T256_DMAW(T256_KT(vW1, dc1), ldsB0 + (PB) + 1024)      \
            if constexpr (TS == 1) { T256_DMAW(T256_KT(vW4, dc0), ldsB0 + (PB) + 2048) T256_DMAW(T256_KT(vW5, dc1), ldsB0 + (PB) + 3072) } \
        }                                                                                                     \
    }
#define T256_ISSUE_B1(PB, F)                                                                                  \
    if ((F) || l_active) {                                                                                    \
        if ((F) || ld_kcb + 128 <= kK_b) {                                                                    \
            T256_DMAW(vW2, ldsB1 + (PB)) T256_DMAW(vW3, ldsB1 + (PB) + 1024)                                  \
            if constexpr (TS == 1) { T256_DMAW(vW6, ldsB1 + (PB) + 2048) T256_DMAW(vW7, ldsB1 + (PB) + 3072) } \
        } else {                                                                                              \
            T256_DMAW(T256_KT(vW2, dc0), ldsB1 + (PB)) T256_DMAW(T256_KT(vW3, dc1), ldsB1 + (PB) + 1024)      \
            if constexpr (TS == 1) { T256_DMAW(T256_KT(vW6, dc0), ldsB1 + (PB) + 2048) T256_DMAW(T256_KT(vW7, dc1), ldsB1 + (PB) + 3072) } \
        }                                                                                                     \
    }
    // move the load cursor to the next K-tile of the stream
#define T256_ADVANCE(F)                                                                                       \
    if ((F) || l_active) {                                                                                    \
        ++l_kt;                                                                                               \
        if ((F) || l_kt < l_kt_end) {                                                                         \
            ++ld_j;                                                                                           \
            if constexpr (C2D) {                                                                              \
                if (++ld_jw == p.cv_kw) { ld_jw = 0; ++ld_jh; }                                               \
                if (ld_j == p.taps) { ld_j = 0; ld_jh = 0; ld_jw = 0; ld_kcb += 128; }                        \
                sA = (ld_jh * p.cv_W + ld_jw) * lda_b + ld_kcb; sW = w0_b + ld_j * wstep_b + ld_kcb;          \
            } else {                                                                                          \
                if (ld_j == p.taps) { ld_j = 0; ld_kcb += 128; }                                              \
                sA = ld_j * lda_b + ld_kcb; sW = ld_j * wts_b + ld_kcb;                                       \
            }                                                                                                 \
        } else {                                                                                              \
            li += nbx;                                                                                        \
            if (li < it_hi) T256_SETUP_ITEM() else l_active = false;                                          \
        }                                                                                                     \
    }

    // ---------------- fragment addressing ----------------
    const int q = lane >> 4, lr = lane & 15;
    const int swz = lr >> 1;
    const uint32_t smem_b = (uint32_t)(uintptr_t)(t256_lds_t*)smem;
    const int fa_row = (TS ? 0 : g * 128) + lr, fb_row = (TS ? wave * 64 : wq * 64) + lr;   // TS = 1: every wave reads all 128 rows
    const uint32_t fA0 = smem_b + fa_row * 128 + ((q ^ swz) << 4);                        // k sub-step 0
    const uint32_t fA1 = smem_b + fa_row * 128 + (((4 + q) ^ swz) << 4);                  // k sub-step 1
    const uint32_t fB0 = smem_b + WOFF + fb_row * 128 + ((q ^ swz) << 4);
    const uint32_t fB1 = smem_b + WOFF + fb_row * 128 + (((4 + q) ^ swz) << 4);
    const uint32_t fA0n = fA0 + BUFB, fA1n = fA1 + BUFB, fB0n = fB0 + BUFB, fB1n = fB1 + BUFB;

    t256_f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = (t256_f4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa00, fa01, fa10, fa11, fa20, fa21, fa30, fa31;     // activation fragments [row tile 0..3][k sub-step]
    bf16x8 fb00, fb01, fb10, fb11;                             // weight fragments, columns 0-31: [col tile][k sub-step]
    bf16x8 fc00, fc01, fc10, fc11;                             // weight fragments, columns 32-63

#ifdef T256_ABL_NOREAD      // timing-only ablation builds (tests/micro): results are wrong by construction
#define T256_DSR(DST, ADDR, IMM) asm volatile("; no read %0 %1" : "=v"(DST) : "v"(ADDR));
#else
#define T256_DSR(DST, ADDR, IMM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(IMM));
#endif
#define T256_READ_A(A0_, A1_, BASE)                                                                           \
    {                                                                                                         \
        T256_DSR(fa00, A0_, (BASE) + 0) T256_DSR(fa01, A1_, (BASE) + 0)                                       \
        T256_DSR(fa10, A0_, (BASE) + 2048) T256_DSR(fa11, A1_, (BASE) + 2048)                                 \
        T256_DSR(fa20, A0_, (BASE) + 4096) T256_DSR(fa21, A1_, (BASE) + 4096)                                 \
        T256_DSR(fa30, A0_, (BASE) + 6144) T256_DSR(fa31, A1_, (BASE) + 6144)                                 \
    }
#define T256_READ_B(X, B0_, B1_, BASE)                                                                        \
    {                                                                                                         \
        T256_DSR(X##00, B0_, (BASE) + 0) T256_DSR(X##01, B1_, (BASE) + 0)                                     \
        T256_DSR(X##10, B0_, (BASE) + 2048) T256_DSR(X##11, B1_, (BASE) + 2048)                               \
    }
#define T256_WAIT_A() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa00), "+v"(fa01), "+v"(fa10), "+v"(fa11), "+v"(fa20), "+v"(fa21), "+v"(fa30), "+v"(fa31));
#define T256_WAIT_B(X) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X##00), "+v"(X##01), "+v"(X##10), "+v"(X##11));
#ifdef T256_ABL_NOMFMA
#define T256_MMA(I, N, X, NI, FA, S) asm volatile("; no mfma" : "+v"(acc[I][N]) : "v"(X##NI##S), "v"(FA##S));
#else
#define T256_MMA(I, N, X, NI, FA, S) acc[I][N] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X##NI##S, FA##S, acc[I][N], 0, 0, 0);
#endif
    // 16 MFMAs: row tiles R0..R0+3 x column tiles C0, C0+1 (weights X) x 2 k sub-steps
#define T256_MMA16(R0, C0, X)                                                                                 \
    {                                                                                                         \
        if (T256_SETPRIO) __builtin_amdgcn_s_setprio(1);                                                      \
        T256_MMA(R0 + 0, C0 + 0, X, 0, fa0, 0) T256_MMA(R0 + 0, C0 + 1, X, 1, fa0, 0)                         \
        T256_MMA(R0 + 1, C0 + 0, X, 0, fa1, 0) T256_MMA(R0 + 1, C0 + 1, X, 1, fa1, 0)                         \
        T256_MMA(R0 + 2, C0 + 0, X, 0, fa2, 0) T256_MMA(R0 + 2, C0 + 1, X, 1, fa2, 0)                         \
        T256_MMA(R0 + 3, C0 + 0, X, 0, fa3, 0) T256_MMA(R0 + 3, C0 + 1, X, 1, fa3, 0)                         \
        T256_MMA(R0 + 0, C0 + 0, X, 0, fa0, 1) T256_MMA(R0 + 0, C0 + 1, X, 1, fa0, 1)                         \
        T256_MMA(R0 + 1, C0 + 0, X, 0, fa1, 1) T256_MMA(R0 + 1, C0 + 1, X, 1, fa1, 1)                         \
        T256_MMA(R0 + 2, C0 + 0, X, 0, fa2, 1) T256_MMA(R0 + 2, C0 + 1, X, 1, fa2, 1)                         \
        T256_MMA(R0 + 3, C0 + 0, X, 0, fa3, 1) T256_MMA(R0 + 3, C0 + 1, X, 1, fa3, 1)                         \
        if (T256_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    // end of an L section: my DMAs older than the last four sections have landed; publish.
    // wmode 0: steady state; 1: the K-tile after an epilogue (its stores and the bias load sit in the queue too);
    // 2: the DMA stream has ended (sections may have issued nothing: drain)
#define T256_LEND(F)                                                                                          \
    {                                                                                                         \
        if (T256_ABL_NOBAR_ && (F)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VMW) : "memory");                \
        else if ((F) || wmode == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(VMW) : "memory");    \
        else if (wmode == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(VMW + NST) : "memory");   \
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    // one K-tile: parity buffer PB is consumed; quarters of the stream's next K-tiles are issued into PN (B1, A1: the K-tile
    // the cursor points at) and, after the cursor has moved, into PB (A0, B0).  A quarter is refilled two sections after the
    // section that read it, so the barrier of the section in between orders the reads before the DMA for every wave.
#if T256_FINEWAIT
    // reads ordered by first use; every MFMA pair waits only for the fragments it consumes (LDS returns in order)
#define T256_W3(N, X, Y, Z) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(X), "+v"(Y), "+v"(Z));
#define T256_W1(N, X) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(X));
#define T256_W2(N, X, Y) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(X), "+v"(Y));
#define T256_MM2(R, C0, X, FA, S) T256_MMA(R, C0 + 0, X, 0, FA, S) T256_MMA(R, C0 + 1, X, 1, FA, S)
#define T256_SEC1_READ(A0P, A1P, B0P, B1P)                                                                    \
        T256_DSR(fb00, B0P, 0) T256_DSR(fb10, B0P, 2048) T256_DSR(fa00, A0P, 0)                               \
        T256_DSR(fa10, A0P, 2048) T256_DSR(fa20, A0P, 4096) T256_DSR(fa30, A0P, 6144)                         \
        T256_DSR(fb01, B1P, 0) T256_DSR(fb11, B1P, 2048) T256_DSR(fa01, A1P, 0)                               \
        T256_DSR(fa11, A1P, 2048) T256_DSR(fa21, A1P, 4096) T256_DSR(fa31, A1P, 6144)
#define T256_SEC1_MMA()                                                                                       \
        if (T256_SETPRIO) __builtin_amdgcn_s_setprio(1);                                                      \
        T256_W3(9, fb00, fb10, fa00) __builtin_amdgcn_sched_barrier(0); T256_MM2(0, 0, fb, fa0, 0)            \
        T256_W1(8, fa10) __builtin_amdgcn_sched_barrier(0); T256_MM2(1, 0, fb, fa1, 0)                        \
        T256_W1(7, fa20) __builtin_amdgcn_sched_barrier(0); T256_MM2(2, 0, fb, fa2, 0)                        \
        T256_W1(6, fa30) __builtin_amdgcn_sched_barrier(0); T256_MM2(3, 0, fb, fa3, 0)                        \
        T256_W3(3, fb01, fb11, fa01) __builtin_amdgcn_sched_barrier(0); T256_MM2(0, 0, fb, fa0, 1)            \
        T256_W1(2, fa11) __builtin_amdgcn_sched_barrier(0); T256_MM2(1, 0, fb, fa1, 1)                        \
        T256_W1(1, fa21) __builtin_amdgcn_sched_barrier(0); T256_MM2(2, 0, fb, fa2, 1)                        \
        T256_W1(0, fa31) __builtin_amdgcn_sched_barrier(0); T256_MM2(3, 0, fb, fa3, 1)                        \
        if (T256_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0);
#define T256_SEC2_READ(B0P, B1P)                                                                              \
        T256_DSR(fc00, B0P, 4096) T256_DSR(fc10, B0P, 4096 + 2048) T256_DSR(fc01, B1P, 4096) T256_DSR(fc11, B1P, 4096 + 2048)
#define T256_SEC2_MMA()                                                                                       \
        if (T256_SETPRIO) __builtin_amdgcn_s_setprio(1);                                                      \
        T256_W2(2, fc00, fc10) __builtin_amdgcn_sched_barrier(0);                                             \
        T256_MM2(0, 2, fc, fa0, 0) T256_MM2(1, 2, fc, fa1, 0) T256_MM2(2, 2, fc, fa2, 0) T256_MM2(3, 2, fc, fa3, 0) \
        T256_W2(0, fc01, fc11) __builtin_amdgcn_sched_barrier(0);                                             \
        T256_MM2(0, 2, fc, fa0, 1) T256_MM2(1, 2, fc, fa1, 1) T256_MM2(2, 2, fc, fa2, 1) T256_MM2(3, 2, fc, fa3, 1) \
        if (T256_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0);
#define T256_SEC3_READ(A0P, A1P)                                                                              \
        T256_DSR(fa00, A0P, 8192) T256_DSR(fa10, A0P, 8192 + 2048) T256_DSR(fa20, A0P, 8192 + 4096) T256_DSR(fa30, A0P, 8192 + 6144) \
        T256_DSR(fa01, A1P, 8192) T256_DSR(fa11, A1P, 8192 + 2048) T256_DSR(fa21, A1P, 8192 + 4096) T256_DSR(fa31, A1P, 8192 + 6144)
#define T256_SEC3_MMA()                                                                                       \
        if (T256_SETPRIO) __builtin_amdgcn_s_setprio(1);                                                      \
        T256_W1(7, fa00) __builtin_amdgcn_sched_barrier(0); T256_MM2(4, 2, fc, fa0, 0)                        \
        T256_W1(6, fa10) __builtin_amdgcn_sched_barrier(0); T256_MM2(5, 2, fc, fa1, 0)                        \
        T256_W1(5, fa20) __builtin_amdgcn_sched_barrier(0); T256_MM2(6, 2, fc, fa2, 0)                        \
        T256_W1(4, fa30) __builtin_amdgcn_sched_barrier(0); T256_MM2(7, 2, fc, fa3, 0)                        \
        T256_W1(3, fa01) __builtin_amdgcn_sched_barrier(0); T256_MM2(4, 2, fc, fa0, 1)                        \
        T256_W1(2, fa11) __builtin_amdgcn_sched_barrier(0); T256_MM2(5, 2, fc, fa1, 1)                        \
        T256_W1(1, fa21) __builtin_amdgcn_sched_barrier(0); T256_MM2(6, 2, fc, fa2, 1)                        \
        T256_W1(0, fa31) __builtin_amdgcn_sched_barrier(0); T256_MM2(7, 2, fc, fa3, 1)                        \
        if (T256_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0);
#else
#define T256_SEC1_READ(A0P, A1P, B0P, B1P) T256_READ_A(A0P, A1P, 0) T256_READ_B(fb, B0P, B1P, 0)
#define T256_SEC1_MMA() T256_WAIT_A() T256_WAIT_B(fb) __builtin_amdgcn_sched_barrier(0); T256_MMA16(0, 0, fb)
#define T256_SEC2_READ(B0P, B1P) T256_READ_B(fc, B0P, B1P, 4096)
#define T256_SEC2_MMA() T256_WAIT_B(fc) __builtin_amdgcn_sched_barrier(0); T256_MMA16(0, 2, fc)
#define T256_SEC3_READ(A0P, A1P) T256_READ_A(A0P, A1P, 8192)
#define T256_SEC3_MMA() T256_WAIT_A() __builtin_amdgcn_sched_barrier(0); T256_MMA16(4, 2, fc)
#endif
#define T256_KTILE(A0P, A1P, B0P, B1P, PB, PN, F)                                                             \
    {                                                                                                         \
        T256_SEC1_READ(A0P, A1P, B0P, B1P)                                                                    \
        T256_ISSUE_B1(PN, F)                                                                                  \
        if (early) T256_LEND(F)                                                                               \
        T256_SEC1_MMA()                                                                                       \
        if (!early) T256_LEND(F)                                                                              \
        T256_SEC2_READ(B0P, B1P)                                                                              \
        T256_ISSUE_A1(PN, F)                                                                                  \
        if (early) T256_LEND(F)                                                                               \
        T256_SEC2_MMA()                                                                                       \
        if (!early) T256_LEND(F)                                                                              \
        T256_SEC3_READ(A0P, A1P)                                                                              \
        T256_ADVANCE(F)                                                                                       \
        if (!(F) && !l_active) wmode = 2;                                                                     \
        T256_ISSUE_A0(PB, F)                                                                                  \
        if (early) T256_LEND(F)                                                                               \
        T256_SEC3_MMA()                                                                                       \
        if (!early) T256_LEND(F)                                                                              \
        T256_ISSUE_B0(PB, F)                                                                                  \
        if (early) T256_LEND(F)                                                                               \
        T256_MMA16(4, 0, fb)                                                                                  \
        if (!early) T256_LEND(F)                                                                              \
    }
    // The two row halves run HALF A SECTION OUT OF PHASE on each SIMD: waves 0-3 ("early") put the section barrier between
    // their reads and their MFMAs (the MFMAs of section k run in the interval after barrier k, beside the other half's reads),
    // waves 4-7 put it after their MFMAs (reads, wait, MFMAs inside one interval, beside the early half's MFMAs and reads).
    // Every wave still executes one barrier per section, each refill still comes two barriers after the reads it overwrites,
    // and each read still comes after the barrier that follows the counted wait of every wave that issued its DMA.
    const bool early = !T256_STAGGER || g == 0;          // TS = 1: waves 0-3 / 4-7 (g is wave >> 2 there as well)

    // ---------------- compute cursor ----------------
    int ci = it_lo + jb;
    int c_z, c_m0, c_n0, c_nkt, c_tile;
#define T256_DECODE_C()                                                                                       \
    {                                                                                                         \
        int tm_, tn_;                                                                                         \
        T256_TILE_OF(ci, c_z, tm_, tn_)                                                                       \
        c_m0 = tm_ << TMS; c_n0 = tn_ << TNS; c_tile = tn_ * tiles_m + tm_;                                   \
        c_nkt = T256_UNI((int)((long)total_kt * (c_z + 1) / p.splitk) - (int)((long)total_kt * c_z / p.splitk)); \
    }
    T256_DECODE_C()
    // bias of this wave's 64 columns, one per lane; loaded by inline asm so that the compiler's waitcnt bookkeeping does not
    // drain the DMA ring for it.  It is older than every DMA of its item and every item has >= 4 K-tiles, so the counted
    // waits of the K loop have retired it long before the epilogue reads it.
    float bias_lane = 0.f;
    const bool has_bias = (OUT == 0 || p.splitk == 1) && p.bias != nullptr;
#define T256_LOAD_BIAS(N0)                                                                                    \
    {                                                                                                         \
        const int col_ = (N0) + (TS ? wave : wq) * 64 + lane;                                                 \
        const float* bp_ = has_bias ? p.bias + (col_ < p.N ? col_ : 0) : reinterpret_cast<const float*>(p.W); \
        asm volatile("global_load_dword %0, %1, off" : "=v"(bias_lane) : "v"(bp_) : "memory");                \
    }

    // ---------------- prologue: K-tile 0 entirely, the A0 / B0 quarters of K-tile 1 ----------------
    T256_LOAD_BIAS(c_n0)
    T256_SETUP_ITEM()
    T256_ISSUE_A0(0, 0) T256_ISSUE_B0(0, 0) T256_ISSUE_B1(0, 0) T256_ISSUE_A1(0, 0)
    T256_ADVANCE(0)
    T256_ISSUE_A0(BUFB, 0) T256_ISSUE_B0(BUFB, 0)
    // K-tile 0's A0 / B0 must have landed: everything but the (up to) VMW youngest DMAs
    if (l_active) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(VMW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    int kt = 0;
    uint32_t pb = 0;                 // parity buffer of the K-tile being consumed
    int wmode = l_active ? 0 : 2;
    for (;;) {
        // fast path: two K-tiles with compile-time parity while neither the compute cursor (last K-tile of its item) nor the load
        // cursor (item change, K tail, end of the stream) meets a boundary
        while (pb == 0 && wmode == 0 && kt + 2 < c_nkt && l_kt + 2 < l_kt_end && ld_kcb + 384 <= kK_b) {
            T256_KTILE(fA0, fA1, fB0, fB1, 0, BUFB, 1)
            T256_KTILE(fA0n, fA1n, fB0n, fB1n, BUFB, 0, 1)
            kt += 2;
        }
        {
            const uint32_t pn = (uint32_t)BUFB - pb;
            const uint32_t a0_ = fA0 + pb, a1_ = fA1 + pb, b0_ = fB0 + pb, b1_ = fB1 + pb;
            T256_KTILE(a0_, a1_, b0_, b1_, pb, pn, 0)
            ++kt;
            pb = pn;
        }
        if (kt == c_nkt) {
            // ================= epilogue of item ci =================
            const int mw = c_m0 + (TS ? 0 : g * 128), nw = c_n0 + (TS ? wave : wq) * 64;
            asm volatile("" : "+v"(bias_lane));
            float bv[4][4];          // bias of the lane's 16 columns nt*16 + 4q + r
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[nt][r] = has_bias ? __shfl(bias_lane, nt * 16 + q * 4 + r, 64) : 0.f;
            if constexpr (OUT == 1) {
                // fp32: split-K slab (raw sums) or final fp32 output.  Buffer stores: always issued (the counted waits after the
                // epilogue rely on it), rows >= M / columns >= N are dropped by the range check.
                const bool slab = p.splitk > 1;
                float* outp = slab ? p.partial + (long)c_z * p.M * p.N : reinterpret_cast<float*>(p.C);
                const int ldo = slab ? p.N : (int)p.ldc;
                const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)(((long)(p.M - 1) * ldo + p.N) * 4), 0x00020000);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = mw + i * 16 + lr;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int col = nw + nt * 16 + q * 4;
                        const bool ok = row < p.M && col < p.N;
                        t256_f4 v = acc[i][nt];
                        acc[i][nt] = (t256_f4){0.f, 0.f, 0.f, 0.f};
                        if (!slab) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = v[r] * sc + bv[nt][r];
                            if (p.addend) {
                                if (ok) {
                                    const bf16_t* ad = reinterpret_cast<const bf16_t*>(p.addend) + (long)row * p.ldadd + col;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] += (float)ad[r];
                                }
                            }
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(t256_u4, v), rsC,
                                                               ok ? (uint32_t)(((long)row * ldo + col) * 4) : T256_OOB, 0, 0);
                    }
                }
            } else {
                const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((long)(p.M - 1) * p.ldc + p.N) * 2), 0x00020000);
                const bf16_t* addp = reinterpret_cast<const bf16_t*>(p.addend);
                // statistics: rows of the next sample (>= rb) / columns of the next group (>= cb); Tlen >= 128, Cg >= 64, Cg % 4 == 0
                const bool st = p.gn_part != nullptr;
                const int rb = st ? (mw / p.Tlen + 1) * p.Tlen : 0x7fffffff;
                const int cb = st ? (nw / p.gn_Cg + 1) * p.gn_Cg : 0x7fffffff;
                float sA1 = 0.f, sA2 = 0.f, sR1 = 0.f, sR2 = 0.f, sC1 = 0.f, sC2 = 0.f, sB1 = 0.f, sB2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = mw + i * 16 + lr;
                    float rA1 = 0.f, rA2 = 0.f, rC1 = 0.f, rC2 = 0.f;
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const int ne = 2 * pr, no = 2 * pr + 1;
                        const uint32_t x0 = t256_pack2(acc[i][ne][0] * sc + bv[ne][0], acc[i][ne][1] * sc + bv[ne][1]);
                        const uint32_t x1 = t256_pack2(acc[i][ne][2] * sc + bv[ne][2], acc[i][ne][3] * sc + bv[ne][3]);
                        const uint32_t y0 = t256_pack2(acc[i][no][0] * sc + bv[no][0], acc[i][no][1] * sc + bv[no][1]);
                        const uint32_t y1 = t256_pack2(acc[i][no][2] * sc + bv[no][2], acc[i][no][3] * sc + bv[no][3]);
                        acc[i][ne] = (t256_f4){0.f, 0.f, 0.f, 0.f};
                        acc[i][no] = (t256_f4){0.f, 0.f, 0.f, 0.f};
                        // lanes 16-31 / 48-63 of x <-> lanes 0-15 / 32-47 of y: even q ends with the whole 8-column chunk of
                        // column tile ne, odd q with the chunk of column tile no
                        const t256_u2 s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                        const t256_u2 s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                        t256_u4 ch = {s0[0], s1[0], s0[1], s1[1]};
                        const int col = nw + (2 * pr + (q & 1)) * 16 + (q >> 1) * 8;
                        const bool ok = row < p.M && col < p.N;
                        if (addp) {
                            if (ok) {
                                const t256_u4 ad = *reinterpret_cast<const t256_u4*>(addp + (long)row * p.ldadd + col);
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    ch[e] = t256_pack2(t256_lo(ch[e]) + t256_lo(ad[e]), t256_hi(ch[e]) + t256_hi(ad[e]));
                            }
                        }
#ifdef T256_ABL_CONTIG      // timing only: every store instruction writes 1 KiB of consecutive bytes
                        __builtin_amdgcn_raw_buffer_store_b128(ch, rsC, (uint32_t)((((long)mw * p.ldc + nw) * 2 & ~1023L) + ((i * 2 + pr) * 8 + wave) * 1024 + lane * 16), 0, 0);
#else
                        if (!T256_ABL_NOSTORE_) __builtin_amdgcn_raw_buffer_store_b128(ch, rsC, ok ? (uint32_t)(((long)row * p.ldc + col) * 2) : T256_OOB, 0, (STRM & 2) ? 16 : 0);
#endif
                        if (st) {
                            // the chunk's two 4-column halves may lie in different groups (Cg % 4 == 0)
                            const float l0 = t256_lo(ch[0]), h0 = t256_hi(ch[0]), l1 = t256_lo(ch[1]), h1 = t256_hi(ch[1]);
                            const float l2 = t256_lo(ch[2]), h2 = t256_hi(ch[2]), l3 = t256_lo(ch[3]), h3 = t256_hi(ch[3]);
                            float u1 = (l0 + h0) + (l1 + h1), u2 = (l0 * l0 + h0 * h0) + (l1 * l1 + h1 * h1);
                            float w1 = (l2 + h2) + (l3 + h3), w2 = (l2 * l2 + h2 * h2) + (l3 * l3 + h3 * h3);
                            if (!ok) { u1 = 0.f; u2 = 0.f; w1 = 0.f; w2 = 0.f; }
                            const float hcu = col >= cb ? 1.f : 0.f, hcw = col + 4 >= cb ? 1.f : 0.f;
                            rA1 += u1 + w1; rA2 += u2 + w2;
                            rC1 += hcu * u1 + hcw * w1; rC2 += hcu * u2 + hcw * w2;
                        }
                    }
                    if (st) {
                        const float hr = row >= rb ? 1.f : 0.f;
                        sA1 += rA1; sA2 += rA2; sC1 += rC1; sC2 += rC2;
                        sR1 += hr * rA1; sR2 += hr * rA2; sB1 += hr * rC1; sB2 += hr * rC2;
                    }
                }
                if (st) {
                    float vals[8] = {sA1, sA2, sR1, sR2, sC1, sC2, sB1, sB2};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float x = vals[k];
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
                        vals[k] = x;
                    }
                    if (lane == 0) {
                        float* dst = p.gn_part + ((long)c_tile * 8 + wave) * 8;    // statistics need split-K 1: one item per tile
                        *reinterpret_cast<t256_f4*>(dst) = (t256_f4){vals[0], vals[1], vals[2], vals[3]};
                        *reinterpret_cast<t256_f4*>(dst + 4) = (t256_f4){vals[4], vals[5], vals[6], vals[7]};
                    }
                }
            }
            // next item (its bias load joins the stores in the queue)
            ci += nbx;
            const bool more = ci < it_hi;
            if (more) T256_DECODE_C()
            T256_LOAD_BIAS(c_n0)
            if (wmode != 2) wmode = 1;
            __builtin_amdgcn_sched_barrier(0);
            if (!more) break;
            kt = 0;
        } else if (wmode == 1) {
            wmode = 0;               // the K-tile after an epilogue is over
        }
    }
}

// Fixed-order reduction of the per-(item, wave) statistics partials into GroupNorm sums [sample][group][2] (fp64).
// One 64-thread block per (sample, group); each thread walks a fixed subset of the candidate entries.
// ts: tile shape of the launch that wrote the partials (0: 256 x 256, wave (g, wq) = 128 rows x 64 columns at (g*128, wq*64);
// 1: 128 x 512, wave w = 128 rows x 64 columns at (0, w*64))
__global__ __launch_bounds__(64) void t256_stats_finalize_kernel(const float* part, double* sums, int M, int N, int Tlen, int Cg, int G, int ts) {
    const int b = blockIdx.x / G, gi = blockIdx.x - b * G;
    const int tms = ts ? 7 : 8, tns = ts ? 9 : 8;
    const int tiles_m = (M + (1 << tms) - 1) >> tms;
    const int r_lo = b * Tlen, r_hi = min((b + 1) * Tlen, M);          // rows of this sample
    const int c_lo = gi * Cg, c_hi = (gi + 1) * Cg;
    const int tm_lo = r_lo >> tms, tm_hi = (r_hi - 1) >> tms;
    const int tn_lo = c_lo >> tns, tn_hi = (c_hi - 1) >> tns;
    const int n_tm = tm_hi - tm_lo + 1, n_tn = tn_hi - tn_lo + 1;
    const int total = n_tm * n_tn * 8;
    double s1 = 0.0, s2 = 0.0;
    for (int e = threadIdx.x; e < total; e += 64) {
        const int wave = e & 7;
        const int t = e >> 3;
        const int tn = tn_lo + t / n_tm, tm = tm_lo + t % n_tm;
        const int item = tn * tiles_m + tm;                            // split-K 1: item index = tile index
        const int mw = ts ? (tm << 7) : (tm << 8) + (wave >> 2) * 128, nw = ts ? (tn << 9) + wave * 64 : (tn << 8) + (wave & 3) * 64;
        const int rb = (mw / Tlen + 1) * Tlen, cb = (nw / Cg + 1) * Cg;
        const float* v = part + ((long)item * 8 + wave) * 8;
        const double A1 = v[0], A2 = v[1], R1 = v[2], R2 = v[3], C1 = v[4], C2 = v[5], B1 = v[6], B2 = v[7];
        // which quadrant of the entry (rows below / from rb) x (columns below / from cb) is (b, gi)?
        const int rlow = mw / Tlen, clow = nw / Cg;
        const int rsel = b == rlow ? 0 : (b == rlow + 1 ? 1 : -1);
        const int csel = gi == clow ? 0 : (gi == clow + 1 ? 1 : -1);
        if (rsel < 0 || csel < 0) continue;
        double q1, q2;
        if (rsel == 0 && csel == 0) { q1 = A1 - R1 - C1 + B1; q2 = A2 - R2 - C2 + B2; }
        else if (rsel == 1 && csel == 0) { q1 = R1 - B1; q2 = R2 - B2; }
        else if (rsel == 0 && csel == 1) { q1 = C1 - B1; q2 = C2 - B2; }
        else { q1 = B1; q2 = B2; }
        s1 += q1; s2 += q2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (threadIdx.x == 0) { sums[(long)blockIdx.x * 2] = s1; sums[(long)blockIdx.x * 2 + 1] = s2; }
}

// split-K combine for the 256 kernel: out = scale * sum_z partial[z] + bias + addend (bf16 or fp32 output), fixed order
template <bool OUT_F32>
__global__ __launch_bounds__(256) void t256_reduce_kernel(const GemmNT p) {
    const long total = (long)p.M * p.N, quads = total >> 2;
    const float sc = p.scale ? *p.scale : 1.0f;
    const int nq = p.N >> 2;
    for (long qd = (long)blockIdx.x * 256 + threadIdx.x; qd < quads; qd += (long)gridDim.x * 256) {
        const int row = (int)(qd / nq), col = (int)(qd - (long)row * nq) * 4;
        float4 v = *reinterpret_cast<const float4*>(p.partial + qd * 4);
        for (int z = 1; z < p.splitk; ++z) {
            const float4 w = *reinterpret_cast<const float4*>(p.partial + (long)z * total + qd * 4);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        float o[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
        if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
            o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w;
        }
        if (OUT_F32) {
            if (p.addend) {
                const bf16_t* ad = reinterpret_cast<const bf16_t*>(p.addend) + (long)row * p.ldadd + col;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += (float)ad[e];
            }
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)row * p.ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            // same rounding sequence as the direct epilogue: round, then add the addend and round again
            bf16_t r[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (bf16_t)o[e];
            if (p.addend) {
                const bf16_t* ad = reinterpret_cast<const bf16_t*>(p.addend) + (long)row * p.ldadd + col;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = (bf16_t)((float)r[e] + (float)ad[e]);
            }
            bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + (long)row * p.ldc + col;
            typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
            bf4 pk; pk[0] = r[0]; pk[1] = r[1]; pk[2] = r[2]; pk[3] = r[3];
            *reinterpret_cast<bf4*>(dst) = pk;
        }
    }
}

// =========================================================================================
// host side
// =========================================================================================
static inline int t256_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

bool gemm_nt256_eligible(int dtype, const GemmNT& p) {
    static const int on = getenv("SGV_GEMM_T256") ? atoi(getenv("SGV_GEMM_T256")) : 1;
    if (!on || dtype != 1) return false;
    if (p.N < 256 || p.M < (p.ts > 0 ? 128 : 256)) return false;      // the 128 x 512 tile shape takes a single 128-row tile too
    if (p.K % 8 || p.N % 8 || p.lda % 8 || p.ldw % 8 || p.w_tap_stride % 8 || p.ldc % 8) return false;
    if (p.addend && p.ldadd % 8) return false;
    if (((uintptr_t)p.C & 15) || (p.addend && ((uintptr_t)p.addend & 15))) return false;
    if (p.taps > 24) return false;
    const long total_kt = (long)p.taps * t256_cdiv(p.K, 64);
    return total_kt >= 8;
}

// Work items = 256x256 tiles x split-K slices, spread over 256 persistent workgroups.  Pick the split that minimises
// (rounds of items) x (K-tiles per item + per-item overhead) + the slab combine pass; every slice keeps >= 4 K-tiles.
int gemm_nt256_pick_splitk(int M, int N, int K, int taps) {
    const long tiles = (long)t256_cdiv(M, 256) * t256_cdiv(N, 256);
    const long total_kt = (long)taps * t256_cdiv(K, 64);
    int best = 1;
    double best_cost = 1e30;
    for (int sk = 1; sk <= 32; ++sk) {
        if (sk > 1 && total_kt / sk < 24) break;
        const double rounds = ceil((double)tiles * sk / 256.0);
        const double per = ceil((double)total_kt / sk) + 2.0;                         // K-tiles + epilogue, in K-tile units (~0.9 us)
        double cost = rounds * per;
        if (sk > 1) cost += (2.0 * sk * (double)M * N * 4.0 / 3.5e12) / 0.9e-6 + 6.0; // slab write + combine pass + launch
        if (cost < best_cost * 0.97) { best_cost = cost; best = sk; }
    }
    return best;
}
size_t gemm_nt256_part_floats(int M, int N, int splitk) {       // either tile shape: 64 floats per work item
    const size_t t0 = (size_t)t256_cdiv(M, 256) * t256_cdiv(N, 256), t1 = (size_t)t256_cdiv(M, 128) * t256_cdiv(N, 512);
    return (t0 > t1 ? t0 : t1) * (size_t)(splitk < 1 ? 1 : splitk) * 64;
}

int launch_gemm_nt256(const GemmNT& p, hipStream_t s) {
    if (!gemm_nt256_eligible(1, p)) return -1;
    if (p.splitk > 1 && !p.partial) return -1;
    const long total_kt = (long)p.taps * t256_cdiv(p.K, 64);
    if (p.splitk < 1 || total_kt / p.splitk < 4) return -1;
    if (p.gn_part && (p.splitk != 1 || p.out_f32 || p.Tlen < 128 || p.gn_Cg < 64 || p.gn_Cg % 4 || p.gn_G < 1 || !p.gn_sums)) return -1;
    if (p.add_W > 0) return -1;          // the strided addend is the 128-row kernel's (gemm_nt_plan keeps such a product there)
    const int ts = p.ts ? 1 : 0;         // 128 x 512 tiles: no 2-D taps, no stream policies, at least one full tile width
    if (ts && (p.cv_kw > 0 || p.N < 512 || p.row0)) return -1;
    const int TH = ts ? 128 : 256, TW = ts ? 512 : 256;
    GemmNT q = p;
    const long arows = p.a_rows > p.M ? p.a_rows : p.M;
    q.a_bytes = ((arows - 1) * p.lda + p.K) * 2;
    q.w_bytes = ((long)(p.taps - 1) * p.w_tap_stride + (long)(p.N - 1) * p.ldw + p.K) * 2;
    if (q.a_bytes >= 0x7FFFFFF0L || q.w_bytes >= 0x7FFFFFF0L) return -1;
    // item order and stream policy (see T256_TILE_OF, STRM).  A one-tap product with a short K and many column panels (the recon
    // head: K = 1024, N = 95 008) streams every weight panel once per XCD and 608 MB of output through the XCD's 4 MiB L2; its
    // activation matrix (M x K, 6.5 MB) does not survive there, so the XCDs each take a band of row tiles (<= 3.5 MiB of
    // activations).  SGV_T256_BAND / SGV_T256_STRM override (A/B runs); GemmNT::band / strm < 0 switch either off (tests).
    static const int band_env = getenv("SGV_T256_BAND") ? atoi(getenv("SGV_T256_BAND")) : -1;
    static const int strm_env = getenv("SGV_T256_STRM") ? atoi(getenv("SGV_T256_STRM")) : -1;
    {
        const int tm_all = t256_cdiv(p.M, TH), tn_all = t256_cdiv(p.N, TW);
        const long a_tile_bytes = (long)TH * p.K * 2 * p.taps;
        int band = p.band, strm = p.strm;                                // 0: chosen here; < 0: off (tests)
        if (band == 0 && strm == 0 && !p.cv_kw && p.taps == 1 && p.splitk == 1 && tn_all >= 64 && (long)tm_all * a_tile_bytes > (7L << 19)) {
            band = (int)((7L << 19) / a_tile_bytes);                     // row tiles whose activations fit 3.5 MiB
            if (band < 1) band = 1;
            const int nb = t256_cdiv(tm_all, band);
            band = t256_cdiv(tm_all, nb);                                // equal bands
            strm = T256_STRM_DEFAULT;
        }
        if (band_env >= 0 && p.band == 0) band = band_env;
        if (strm_env >= 0 && p.strm == 0) strm = strm_env;
        if (band < 0 || band >= tm_all) band = 0;
        if (strm < 0) strm = 0;
        q.band = band; q.strm = (!ts && !p.cv_kw && p.taps == 1 && !(p.splitk > 1 || p.out_f32)) ? (strm & 3) : 0;
    }
    const int nitems = t256_cdiv(p.M, TH) * t256_cdiv(p.N, TW) * p.splitk;
    int grid = ((nitems + 7) / 8) * 8;
    if (grid > 256) grid = 256;
    const bool f32 = p.splitk > 1 || p.out_f32;
    if (ts) {
        if (p.taps > 1) {
            if (f32) hipLaunchKernelGGL((gemm_nt_t256_kernel<1, true, false, 0, 1>), dim3(grid), dim3(512), 0, s, q);
            else hipLaunchKernelGGL((gemm_nt_t256_kernel<0, true, false, 0, 1>), dim3(grid), dim3(512), 0, s, q);
        } else {
            if (f32) hipLaunchKernelGGL((gemm_nt_t256_kernel<1, false, false, 0, 1>), dim3(grid), dim3(512), 0, s, q);
            else hipLaunchKernelGGL((gemm_nt_t256_kernel<0, false, false, 0, 1>), dim3(grid), dim3(512), 0, s, q);
        }
    } else if (p.cv_kw > 0) {
        if (p.taps % p.cv_kw || p.cv_S < 1 || p.cv_P < 0 || p.cv_H < 1 || p.cv_W < 1 || p.cv_Ho < 1 || p.cv_Wo < 1) return -1;
        if (p.M % (p.cv_Ho * p.cv_Wo) || p.a_rows != (long)(p.M / (p.cv_Ho * p.cv_Wo)) * p.cv_H * p.cv_W) return -1;
        if (p.gn_part || p.row0) return -1;
        if (f32) hipLaunchKernelGGL((gemm_nt_t256_kernel<1, true, true>), dim3(grid), dim3(512), 0, s, q);
        else hipLaunchKernelGGL((gemm_nt_t256_kernel<0, true, true>), dim3(grid), dim3(512), 0, s, q);
    } else if (p.taps > 1) {
        if (f32) hipLaunchKernelGGL((gemm_nt_t256_kernel<1, true>), dim3(grid), dim3(512), 0, s, q);
        else hipLaunchKernelGGL((gemm_nt_t256_kernel<0, true>), dim3(grid), dim3(512), 0, s, q);
    } else {
        if (f32) hipLaunchKernelGGL((gemm_nt_t256_kernel<1, false>), dim3(grid), dim3(512), 0, s, q);
        else if (q.strm == 1) hipLaunchKernelGGL((gemm_nt_t256_kernel<0, false, false, 1>), dim3(grid), dim3(512), 0, s, q);
        else if (q.strm == 2) hipLaunchKernelGGL((gemm_nt_t256_kernel<0, false, false, 2>), dim3(grid), dim3(512), 0, s, q);
        else if (q.strm == 3) hipLaunchKernelGGL((gemm_nt_t256_kernel<0, false, false, 3>), dim3(grid), dim3(512), 0, s, q);
        else hipLaunchKernelGGL((gemm_nt_t256_kernel<0, false>), dim3(grid), dim3(512), 0, s, q);
    }
    if (p.splitk > 1) {
        int b4 = (int)(((long)p.M * p.N / 4 + 255) / 256);
        if (b4 > 4096) b4 = 4096;
        if (p.out_f32) hipLaunchKernelGGL(t256_reduce_kernel<true>, dim3(b4), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(t256_reduce_kernel<false>, dim3(b4), dim3(256), 0, s, p);
    }
    if (p.gn_part) {
        const int B = t256_cdiv(p.M, p.Tlen);
        hipLaunchKernelGGL(t256_stats_finalize_kernel, dim3(B * p.gn_G), dim3(64), 0, s, p.gn_part, p.gn_sums, p.M, p.N, p.Tlen, p.gn_Cg, p.gn_G, ts);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- kernel choice -------------------------------------------------------------------------------------------------
// costs in units of one K-tile of the 256 kernel (~1.7 us): measured on MI355X with tests/micro/g256_harness.hip
static double t256_cost(int M, int N, long total_kt, int sk, int ts = 0) {
    const double items = (double)t256_cdiv(M, ts ? 128 : 256) * t256_cdiv(N, ts ? 512 : 256) * sk;
    const double rounds = ceil(items / 256.0);
    // a K-tile of the 128 x 512 tile moves 80 KiB instead of 64 and issues 10 DMA pieces per wave instead of 8: measured 11-13 %
    // longer (5120 x 5120 x 5 taps: 630 us in one round of 250 items = the 256 x 256 kernel's 565 us + its 65 us tail)
    static const double wide_kt = getenv("SGV_T256_WIDE_KT") ? atof(getenv("SGV_T256_WIDE_KT")) : 1.13;
    double c = rounds * (ceil((double)total_kt / sk) * (ts ? wide_kt : 1.0) + 2.5) + 3.0;
    if (sk > 1) c += (2.0 * sk * (double)M * N * 4.0 / 3.5e12) / 1.7e-6 + 4.0;
    return c;
}
static int t256_best_sk(int M, int N, long total_kt, size_t partial_floats, double* cost, int ts = 0) {
    int best = 1; double bc = 1e30;
    for (int sk = 1; sk <= 32; ++sk) {
        if (sk > 1 && (total_kt / sk < 24 || (size_t)sk * M * N > partial_floats)) break;
        const double c = t256_cost(M, N, total_kt, sk, ts);
        if (c < bc * 0.97) { bc = c; best = sk; }
    }
    if (cost) *cost = bc;
    return best;
}
GemmPlan gemm_nt_plan(int dtype, const GemmNT& p, size_t partial_floats, int want_stats) {
    static const double min_gf = getenv("SGV_T256_MIN_GF") ? atof(getenv("SGV_T256_MIN_GF")) : 30.0;
    GemmPlan pl = {0, 1, 1, p.M, 0};
    const long total_kt = (long)p.taps * t256_cdiv(p.K, 64);
    const double gf = 2.0e-9 * p.M * p.N * p.K * p.taps;
    const long tiles256 = (long)t256_cdiv(p.M, 256) * t256_cdiv(p.N, 256);
    const bool big = gemm_nt256_eligible(dtype, p) && (p.N >= 1024 || tiles256 >= 200) && gf >= min_gf && !p.out_f32 && total_kt >= 8 && p.add_W <= 0;
    if (!big) {
        pl.sk_main = gemm_nt_pick_splitk(p.M, p.N, p.K, p.taps, dtype);
        if ((size_t)pl.sk_main * p.M * p.N > partial_floats) pl.sk_main = 1;
        pl.fuse_stats = 0;
        return pl;
    }
    const bool stats_ok = want_stats && p.Tlen >= 128 && p.gn_Cg >= 64 && p.gn_Cg % 4 == 0;
    // the 128 x 512 tile (kind 3): all rows in 128-row tiles, no tail launch.  SGV_T256_WIDE: 0 never, 1 where the cost model
    // prefers it (default), 2 wherever it is eligible
    static const int wide = getenv("SGV_T256_WIDE") ? atoi(getenv("SGV_T256_WIDE")) : 1;
    // at least four column tiles (N >= 2048): with two (N = 1024: the K = 95 008 and 5120 -> 1024 layers) every workgroup re-reads
    // half of the weight matrix for 128 rows of output and the tile measured slower than main + tail (70.6 vs 64.5 us, 613 vs 605 us)
    const bool wide_ok = wide && !p.cv_kw && p.N >= (wide == 2 ? 512 : 2048) && p.ts >= 0;
    double c_all, c_main = 1e30, c_wide = 1e30;
    const int sk_all = stats_ok ? 1 : t256_best_sk(p.M, p.N, total_kt, partial_floats, &c_all);
    if (stats_ok) {
        pl.kind = 1; pl.sk_main = 1; pl.fuse_stats = 1;
        if (wide_ok && wide == 2) pl.kind = 3;        // the recon head measured the same either way (592 vs 594 us): it keeps its banded 256 x 256 order
        return pl;
    }
    const int sk_wide = wide_ok ? t256_best_sk(p.M, p.N, total_kt, partial_floats, &c_wide, 1) : 1;
    const int rem = p.M & 255;
    int sk_main = 1, sk_tail = 1;
    if (rem > 0 && rem <= 128 && p.M - rem >= 256 && !p.cv_kw) {
        sk_main = t256_best_sk(p.M - rem, p.N, total_kt, partial_floats, &c_main);
        sk_tail = gemm_nt_pick_splitk(rem, p.N, p.K, p.taps, dtype);
        if ((size_t)sk_tail * rem * p.N > partial_floats) sk_tail = 1;
        // the tail: 128x256 tiles of the gemm.hip kernel at ~60 % of the 256 kernel's rate per CU, + its combine pass and launch
        const double tail_tiles = (double)t256_cdiv(p.N, 256) * sk_tail;
        c_main += ceil(tail_tiles / 256.0) * ((double)total_kt / sk_tail * 0.85 + 4.0) + 6.0;
    }
    if (c_main < c_all) { pl.kind = 2; pl.sk_main = sk_main; pl.sk_tail = sk_tail; pl.m_main = p.M - rem; }
    else { pl.kind = 1; pl.sk_main = sk_all; }
    if (wide_ok && (wide == 2 || c_wide < (c_main < c_all ? c_main : c_all))) { pl.kind = 3; pl.sk_main = sk_wide; pl.sk_tail = 1; pl.m_main = p.M; }
    return pl;
}
int gemm_nt_tail_split(int dtype, const GemmNT& p, const GemmPlan& pl, size_t tail_partial_floats) {
    static const int on = getenv("SGV_TAIL_CONCURRENT") ? atoi(getenv("SGV_TAIL_CONCURRENT")) : 1;
    const int rem = p.M - pl.m_main;
    if (!on || dtype != 1 || pl.kind != 2 || pl.fuse_stats || p.cv_kw || rem != 128 || p.N < 512 || p.row0 || p.out_f32 || p.add_W > 0 || p.trow0) return 0;
    if (pl.m_main < p.pad || p.taps > 24) return 0;
    const int ctiles = (p.N + 511) / 512;
    const long total_kt = (long)p.taps * ((p.K + 63) / 64);
    const int main_items = (pl.m_main / 256) * ((p.N + 255) / 256) * pl.sk_main;
    int sk_t = ctiles <= 16 ? 16 / ctiles : 0;
    while (sk_t > 1 && total_kt / sk_t < 24) --sk_t;
    if (sk_t < 1 || main_items + ctiles * sk_t > 256) return 0;
    // a K-tile of the 128 x 512 tile costs 1.13 of the square tile's (t256_cost): the tail must be over before the main launch
    if ((double)total_kt / sk_t * 1.13 > (double)total_kt / pl.sk_main * 1.05) return 0;
    if ((size_t)sk_t * rem * p.N > tail_partial_floats) return 0;
    return sk_t;
}
int launch_gemm_nt_main(const GemmNT& p, const GemmPlan& pl, hipStream_t s) {
    GemmNT q = p;
    q.splitk = pl.sk_main; q.gn_part = nullptr; q.gn_sums = nullptr; q.ts = 0; q.M = pl.m_main; q.a_rows = p.M;
    return launch_gemm_nt256(q, s);
}
int launch_gemm_nt_tail(const GemmNT& p, const GemmPlan& pl, int sk_tail, float* tail_partial, hipStream_t s) {
    GemmNT t = p;
    const int rem = p.M - pl.m_main;
    t.gn_part = nullptr; t.gn_sums = nullptr;
    t.A = (const char*)p.A + (size_t)pl.m_main * p.lda * 2;
    t.C = (char*)p.C + (size_t)pl.m_main * p.ldc * 2;
    if (p.addend) t.addend = (const char*)p.addend + (size_t)pl.m_main * p.ldadd * 2;
    t.M = rem; t.a_rows = rem; t.trow0 = pl.m_main % p.Tlen; t.ts = 1; t.splitk = sk_tail; t.partial = tail_partial; t.band = -1; t.strm = -1;
    return launch_gemm_nt256(t, s);
}
int launch_gemm_nt_planned(int dtype, const GemmNT& p, const GemmPlan& pl, hipStream_t s) {
    if (pl.kind == 0) {
        GemmNT q = p; q.splitk = pl.sk_main; q.gn_part = nullptr;
        return launch_gemm_nt(dtype, q, s);
    }
    GemmNT q = p;
    q.splitk = pl.sk_main;
    if (!pl.fuse_stats) { q.gn_part = nullptr; q.gn_sums = nullptr; }
    if (pl.kind == 1) { q.ts = 0; return launch_gemm_nt256(q, s); }
    if (pl.kind == 3) { q.ts = 1; return launch_gemm_nt256(q, s); }
    q.ts = 0;
    q.M = pl.m_main; q.a_rows = p.M;
    int r = launch_gemm_nt256(q, s);
    if (r) return r;
    GemmNT t = p;
    t.gn_part = nullptr; t.gn_sums = nullptr;
    t.row0 = pl.m_main; t.splitk = pl.sk_tail;
    return launch_gemm_nt(dtype, t, s);
}
