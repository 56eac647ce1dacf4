// MFMA GEMM kernels for the channels-last conv stacks (gfx950).
//
//  gemm_nt : C[m][n] = scale * sum_{tap,k} A[m + tap - pad][k] * W[tap][n][k] + bias[n] (+ addend[m][n])
//            forward convs (reference Conv1d/ConvTranspose1d call sites: modules/encoder.py:34,43,
//            modules/common.py:84,110,135-141, modules/decoder.py:31,118,135,145,155,164) and their dX.
//            Rows m = b*Tlen + t; a tap that leaves the sample's [0,Tlen) window contributes zero.
//  gemm_tn : dW[tap][n1][n2] = sum_m dY[m][n1] * X[m + tap - pad][n2]   (weight gradients)
//
// Both use 128x128 block tiles, 4 waves (2x2) of 64x64, 32x32 MFMA tiles
// (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32), fp32 accumulation, register-staged
// global->LDS double buffering with one barrier per K step.  16-byte global loads; K tails and
// sample-boundary taps are zero-filled at chunk granularity.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "sgv_common.h"

// minimum waves per SIMD asked of the register allocator for the GEMM kernels: 3 (168 VGPRs; a dozen
// prologue/epilogue spills) measured 3-6 % faster end-to-end than the uncapped 220-VGPR build
#ifndef SGV_GEMM_MIN_WAVES
#define SGV_GEMM_MIN_WAVES 3
#endif

// zero a 16-byte chunk with an integer mask (0 or ~0): a plain AND cannot be turned into a memory select
// XCD-aware block -> work-item map.  Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an
// XCD and its 4 MiB L2), so give XCD x the x-th CONTIGUOUS chunk of the logical order: the ~96 blocks
// an XCD runs concurrently then form a compact patch of the tile grid and share operand panels in L2.
// Bijective for any n (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}
// Grouped raster inside an XCD chunk: patches of 8 tile-rows, row index fastest inside a patch, so the ~96
// blocks an XCD runs at once cover an ~8 x 12 patch and share ~20 operand panels (which fit its 4 MiB L2)
// instead of the ~45 of a thin strip.
__device__ __forceinline__ void grouped_raster(int pid, int tiles_r, int tiles_c, int& tr, int& tc) {
    constexpr int G = 8;
    const int per_group = G * tiles_c;
    const int gid = pid / per_group;
    const int first = gid * G;
    const int gsz = min(tiles_r - first, G);
    const int in_g = pid - gid * per_group;
    tc = in_g / gsz;
    tr = first + (in_g - tc * gsz);
}
__device__ __forceinline__ uint4 mask4(uint4 v, uint32_t m) { return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m); }

// 16-byte buffer load: 32-bit byte offset against a wave-uniform descriptor; offsets at/after num_records
// return zeros from the hardware range check, which is how masked taps / K tails / tile edges are zero-filled
// (no branches, no mask registers).  OOB_OFF is above every operand size the launchers accept.
typedef int v4i32 __attribute__((ext_vector_type(4)));
constexpr uint32_t OOB_OFF = 0x7FFFFFF0u;
__device__ __forceinline__ uint4 bload16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// =========================================================================================
// NT
// =========================================================================================
// C2D: 2-D taps (GemmNT::cv_*): rows are output pixels of a channels-last image batch, a tap is a (kh, kw) window offset
// NW: narrow form for N <= 64 (the conditioner's 16-64 channel layers on a million rows): 128 x 64 tile, the four waves stacked
//     along M (32 rows x 64 columns each) -- half the MFMAs, weight-tile loads, fragment reads and epilogue work of the square tile
template <typename T, int KCH, bool C2D = false, bool NW = false>
__global__ __launch_bounds__(256, SGV_GEMM_MIN_WAVES) void gemm_nt_kernel(const GemmNT p) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = KCH * EPC;
    constexpr bool IS_BF16 = sizeof(T) == 2;
    // LDS row pitch: +16 B (bf16, ds_read_b128 conflict-free: pitch/16 odd) / +4 B (fp32, pitch/4 odd)
    constexpr int ROWB = KCH * 16 + (IS_BF16 ? 16 : 4);
    constexpr int TILEB = 128 * ROWB;
    constexpr int LPT = 128 * KCH / 256;
    constexpr int RSTEP = 256 / KCH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILEB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = NW ? wave : wave >> 1, wn = NW ? 0 : wave & 1;
    constexpr int TN_ = NW ? 64 : 128;           // tile width
    const int tiles_n = (p.N + TN_ - 1) / TN_, tiles_m = (p.M - p.row0 + 127) >> 7;
    const int ntiles = tiles_n * tiles_m;
    // 1-D grid of ntiles*splitk blocks: XCD-chunked, split-K slice slowest, grouped tile raster within.
    const int logical = xcd_remap(blockIdx.x, ntiles * p.splitk);
    const int z = logical / ntiles;
    const int tile = logical - z * ntiles;
    int tm, tn;
    grouped_raster(tile, tiles_m, tiles_n, tm, tn);
    const int m0 = p.row0 + (tm << 7), n0 = tn * TN_;
    const int kchunks = (p.K + BK - 1) / BK;
    const int total = p.taps * kchunks;
    const int s_begin = (int)((long)total * z / p.splitk);
    const int s_end = (int)((long)total * (z + 1) / p.splitk);

    static_assert(LPT == 2, "two 16-byte load slots per operand per thread");
    const int kq = tid % KCH;
    const int r0 = tid / KCH;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (int)p.w_bytes, 0x00020000);
    // named scalars only (no per-thread arrays: hipcc demoted them to scratch memory)
    const int am0 = m0 + r0, am1 = m0 + r0 + RSTEP;
    const int wn0 = n0 + r0, wn1 = n0 + r0 + RSTEP;
    const bool aok0 = am0 < p.M, aok1 = am1 < p.M;
    const bool wok0 = wn0 < p.N, wok1 = !NW && wn1 < p.N;
    constexpr int ESZ = (int)sizeof(T);
    // per-tap validity of this thread's two activation rows as bit masks (taps <= 31)
    uint32_t am0_mask = 0u, am1_mask = 0u;
    uint32_t abase0, abase1;
    if constexpr (C2D) {
        // row -> (b, oh, ow); the base is the window's top-left input pixel (it may lie outside the image: the offset then
        // wraps modulo 2^32 and only the taps whose pixel is inside -- where the sum is the true offset -- are loaded)
        const int hw = p.cv_Ho * p.cv_Wo;
        const int b0_ = am0 / hw, q0_ = am0 - b0_ * hw, b1_ = am1 / hw, q1_ = am1 - b1_ * hw;
        const int oh0 = q0_ / p.cv_Wo, oh1 = q1_ / p.cv_Wo;
        const int ih0 = oh0 * p.cv_S - p.cv_P, iw0 = (q0_ - oh0 * p.cv_Wo) * p.cv_S - p.cv_P;
        const int ih1 = oh1 * p.cv_S - p.cv_P, iw1 = (q1_ - oh1 * p.cv_Wo) * p.cv_S - p.cv_P;
        abase0 = (uint32_t)(((((long)b0_ * p.cv_H + ih0) * p.cv_W + iw0) * p.lda + kq * EPC) * ESZ);
        abase1 = (uint32_t)(((((long)b1_ * p.cv_H + ih1) * p.cv_W + iw1) * p.lda + kq * EPC) * ESZ);
        for (int j = 0, kh = 0, kw = 0; j < p.taps; ++j) {
            if (aok0 && (unsigned)(ih0 + kh) < (unsigned)p.cv_H && (unsigned)(iw0 + kw) < (unsigned)p.cv_W) am0_mask |= 1u << j;
            if (aok1 && (unsigned)(ih1 + kh) < (unsigned)p.cv_H && (unsigned)(iw1 + kw) < (unsigned)p.cv_W) am1_mask |= 1u << j;
            if (++kw == p.cv_kw) { kw = 0; ++kh; }
        }
    } else {
        const int at0 = am0 % p.Tlen, at1 = am1 % p.Tlen;
        abase0 = (uint32_t)(((long)am0 * p.lda + kq * EPC) * ESZ);
        abase1 = (uint32_t)(((long)am1 * p.lda + kq * EPC) * ESZ);
        for (int j = 0; j < p.taps; ++j) {
            if (aok0 && (unsigned)(at0 + j - p.pad) < (unsigned)p.Tlen) am0_mask |= 1u << j;
            if (aok1 && (unsigned)(at1 + j - p.pad) < (unsigned)p.Tlen) am1_mask |= 1u << j;
        }
    }
    const uint32_t wbase0 = wok0 ? (uint32_t)(((long)wn0 * p.ldw + kq * EPC) * ESZ) : OOB_OFF;
    const uint32_t wbase1 = wok1 ? (uint32_t)(((long)wn1 * p.ldw + kq * EPC) * ESZ) : OOB_OFF;
    // two register sets (A, B): loads for tile s+2 are issued while tile s is multiplied and tile s+1 is
    // still in flight (prefetch distance 2; the compiler's in-order vmcnt(4) retires only the older set)
    uint4 ra0A, ra1A, rw0A, rw1A, ra0B, ra1B, rw0B, rw1B;

    // K-step state carried incrementally (channel chunk outer, tap inner): no divisions / 64-bit multiplies in
    // the loop.  An ablation with loads, LDS reads and MFMAs all removed showed the old per-step index
    // arithmetic alone cost ~40 % of the kernel time.
    const int lda_b = (int)(p.lda * ESZ), wts_b = (int)(p.w_tap_stride * ESZ);
    int ld_j = 0, ld_kcb = 0, ld_aoff = 0, ld_woff = 0;     // state of the NEXT tile to load
    int ld_jw = 0;                                           // C2D: kw of tap ld_j
    // C2D: a tap step moves one pixel right, a kernel-row wrap moves to the next image row; the weights advance by
    // +- one tap (cv_flip walks them backwards from the last tap)
    const int a0_b = C2D ? 0 : -p.pad * lda_b;
    const int rowskip_b = C2D ? (p.cv_W - p.cv_kw) * lda_b : 0;
    const int wstep_b = (C2D && p.cv_flip) ? -wts_b : wts_b;
    const int w0_b = (C2D && p.cv_flip) ? (p.taps - 1) * wts_b : 0;
    {
        const int kci0 = s_begin / p.taps;
        ld_j = s_begin - kci0 * p.taps;
        ld_kcb = kci0 * BK * ESZ;
        if constexpr (C2D) {
            const int kh0 = ld_j / p.cv_kw;
            ld_jw = ld_j - kh0 * p.cv_kw;
            ld_aoff = (kh0 * p.cv_W + ld_jw) * lda_b + ld_kcb;
        } else {
            ld_aoff = (ld_j - p.pad) * lda_b + ld_kcb;
        }
        ld_woff = w0_b + ld_j * wstep_b + ld_kcb;
    }
    const int klim_b = (p.K - kq * EPC) * ESZ;               // chunk valid iff ld_kcb < klim_b
#define SGV_NT_GLOAD(S, X)                                                                                    \
    {                                                                                                         \
        const bool kok_ = ld_kcb < klim_b;                                                                    \
        const bool pa0 = kok_ && ((am0_mask >> ld_j) & 1u);                                                   \
        const bool pa1 = kok_ && ((am1_mask >> ld_j) & 1u);                                                   \
        ra0##X = bload16(rsA, pa0 ? abase0 + (uint32_t)ld_aoff : OOB_OFF);                                    \
        ra1##X = bload16(rsA, pa1 ? abase1 + (uint32_t)ld_aoff : OOB_OFF);                                    \
        rw0##X = bload16(rsW, kok_ ? wbase0 + (uint32_t)ld_woff : OOB_OFF);                                   \
        if constexpr (!NW) rw1##X = bload16(rsW, kok_ ? wbase1 + (uint32_t)ld_woff : OOB_OFF);                \
        ++ld_j; ld_aoff += lda_b; ld_woff += wstep_b;                                                         \
        if constexpr (C2D) { if (++ld_jw == p.cv_kw) { ld_jw = 0; ld_aoff += rowskip_b; } }                   \
        if (ld_j == p.taps) { ld_j = 0; ld_jw = 0; ld_kcb += BK * ESZ; ld_aoff = ld_kcb + a0_b; ld_woff = ld_kcb + w0_b; } \
    }
#define SGV_NT_ST1(PTR, V)                                                                                    \
    if constexpr (IS_BF16) { *reinterpret_cast<uint4*>(PTR) = (V); }                                          \
    else { uint32_t* d_ = reinterpret_cast<uint32_t*>(PTR); d_[0] = (V).x; d_[1] = (V).y; d_[2] = (V).z; d_[3] = (V).w; }
#define SGV_NT_SSTORE(BUF, X)                                                                                 \
    {                                                                                                         \
        unsigned char* sa_ = smem + (BUF) * 2 * TILEB + r0 * ROWB + kq * 16;                                  \
        unsigned char* sw_ = sa_ + TILEB;                                                                     \
        SGV_NT_ST1(sa_, ra0##X) SGV_NT_ST1(sa_ + RSTEP * ROWB, ra1##X)                                        \
        SGV_NT_ST1(sw_, rw0##X)                                                                               \
        if constexpr (!NW) { SGV_NT_ST1(sw_ + RSTEP * ROWB, rw1##X) }                                         \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int a_frag_off = (NW ? wm * 32 + lr : wm * 64 + lr) * ROWB;
    const int w_frag_off = (wn * 64 + lr) * ROWB;
#define SGV_NT_COMPUTE(BUF)                                                                                   \
    {                                                                                                         \
        const unsigned char* sa_ = smem + (BUF) * 2 * TILEB + a_frag_off;                                     \
        const unsigned char* sw_ = smem + (BUF) * 2 * TILEB + TILEB + w_frag_off;                             \
        if constexpr (IS_BF16) {                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                          \
                const bf16x8 a0_ = *reinterpret_cast<const bf16x8*>(sa_ + (ks * 2 + lh) * 16);                \
                const bf16x8 b0_ = *reinterpret_cast<const bf16x8*>(sw_ + (ks * 2 + lh) * 16);                \
                const bf16x8 b1_ = *reinterpret_cast<const bf16x8*>(sw_ + 32 * ROWB + (ks * 2 + lh) * 16);    \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b0_, acc[0][0], 0, 0, 0);            \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b1_, acc[0][1], 0, 0, 0);            \
                if constexpr (!NW) {                                                                          \
                    const bf16x8 a1_ = *reinterpret_cast<const bf16x8*>(sa_ + 32 * ROWB + (ks * 2 + lh) * 16); \
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b0_, acc[1][0], 0, 0, 0);        \
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b1_, acc[1][1], 0, 0, 0);        \
                }                                                                                             \
            }                                                                                                 \
        } else {                                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < BK / 2; ++ks) {                                           \
                const float a0_ = *reinterpret_cast<const float*>(sa_ + (ks * 2 + lh) * 4);                   \
                const float a1_ = *reinterpret_cast<const float*>(sa_ + 32 * ROWB + (ks * 2 + lh) * 4);       \
                const float b0_ = *reinterpret_cast<const float*>(sw_ + (ks * 2 + lh) * 4);                   \
                const float b1_ = *reinterpret_cast<const float*>(sw_ + 32 * ROWB + (ks * 2 + lh) * 4);       \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b0_, acc[0][0], 0, 0, 0);               \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b1_, acc[0][1], 0, 0, 0);               \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b0_, acc[1][0], 0, 0, 0);               \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b1_, acc[1][1], 0, 0, 0);               \
            }                                                                                                 \
        }                                                                                                     \
    }

#define SGV_NT_STEP(XL, XS)    /* load tile s+2 into set XL, multiply tile s, stage set XS (tile s+1) */  \
    {                                                                                                         \
        SGV_NT_GLOAD(s + 2, XL);                                                                              \
        __builtin_amdgcn_sched_barrier(0);   /* keep hipcc from sinking the loads below the MFMAs */          \
        SGV_NT_COMPUTE(cur);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        SGV_NT_SSTORE(cur ^ 1, XS);                                                                           \
        __syncthreads();                                                                                      \
        cur ^= 1; ++s;                                                                                        \
    }
    if (s_begin < s_end) {
        int s = s_begin, cur = 0;
        SGV_NT_GLOAD(s, A);
        if (s + 1 < s_end) SGV_NT_GLOAD(s + 1, B);
        SGV_NT_SSTORE(0, A);
        __syncthreads();
        // invariant at loop top: LDS[cur] holds tile s, set B holds tile s+1 (in flight)
        while (s + 3 < s_end) {
            SGV_NT_STEP(A, B)
            SGV_NT_STEP(B, A)
        }
        const int rem = s_end - s;   // 1..3 tiles left
        if (rem == 3) {
            SGV_NT_STEP(A, B)
            SGV_NT_COMPUTE(cur);
            SGV_NT_SSTORE(cur ^ 1, A);
            __syncthreads();
            cur ^= 1;
            SGV_NT_COMPUTE(cur);
        } else if (rem == 2) {
            SGV_NT_COMPUTE(cur);
            SGV_NT_SSTORE(cur ^ 1, B);
            __syncthreads();
            cur ^= 1;
            SGV_NT_COMPUTE(cur);
        } else {
            SGV_NT_COMPUTE(cur);
        }
    }
#undef SGV_NT_STEP
#undef SGV_NT_GLOAD
#undef SGV_NT_SSTORE
#undef SGV_NT_ST1
#undef SGV_NT_COMPUTE

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const float sc = p.scale ? *p.scale : 1.0f;
    const bool full = (m0 + 128 <= p.M) && (n0 + TN_ <= p.N);
    const T* addp = reinterpret_cast<const T*>(p.addend);
    if constexpr (IS_BF16) {
        if (p.splitk == 1 && !p.out_f32) {
            // bf16 output: stage the 128x128 tile through the (now idle) LDS buffers and write whole 256-byte
            // rows with 16-byte stores.  Storing straight from the accumulators costs 64 two-byte stores per
            // lane that each touch half a cache line; on the K=1024 recon-head GEMM that epilogue dominated.
            constexpr int CP = 272;   // LDS row pitch of the C tile (256 B + 16 B pad)
            static_assert(128 * CP <= 4 * TILEB, "C tile must fit in the staging buffers");
            __syncthreads();          // every wave is done reading its last operand tile
#pragma unroll
            for (int a = 0; a < (NW ? 1 : 2); ++a) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int lcol = wn * 64 + b * 32 + lr;
                    const int gcol = n0 + lcol;
                    const float bv = (p.bias && gcol < p.N) ? p.bias[gcol] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lrow = (NW ? wm * 32 : wm * 64 + a * 32) + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        *reinterpret_cast<bf16_t*>(smem + lrow * CP + lcol * 2) = (bf16_t)(acc[a][b][r] * sc + bv);
                    }
                }
            }
            __syncthreads();
            bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C);
            // GroupNorm statistics of the tile as it is stored (p.gn_sums): a 128-row tile meets at most one sample
            // boundary (Tlen >= 128) and a 128-column tile at most one group boundary (gn_Cg >= 128).  Per thread:
            // all / rows of the second sample / columns of the second group / both; the four (sample, group) sums
            // follow by inclusion-exclusion.
            const bool st = p.gn_sums != nullptr;
            const int rb = st ? (m0 / p.Tlen + 1) * p.Tlen : 0x7fffffff;     // first row of the next sample
            const int cb = st ? (n0 / p.gn_Cg + 1) * p.gn_Cg : 0x7fffffff;   // first column of the next group
            const bool csplit = cb < n0 + 128;
            float sA1 = 0.f, sA2 = 0.f, sR1 = 0.f, sR2 = 0.f, sC1 = 0.f, sC2 = 0.f, sB1 = 0.f, sB2 = 0.f;
#pragma unroll
            for (int i = 0; i < (NW ? 4 : 8); ++i) {
                const int c = tid + i * 256;          // 2048 chunks of 8 bf16 (NW: 1024, 8 per row)
                const int lrow = NW ? c >> 3 : c >> 4, lc8 = (NW ? c & 7 : c & 15) * 8;
                const int grow = m0 + lrow, gcol = n0 + lc8;
                if (grow < p.M && gcol < p.N) {       // N % 8 == 0: a chunk is entirely in or out
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + lrow * CP + lc8 * 2);
                    if (addp) {
                        // add_W > 0: the addend is a half-resolution image batch [B][ceil(add_H/2)][ceil(add_W/2)][ldadd] added at the
                        // even pixels of this output's [B][add_H][add_W] rows only (input gradient of a stride-2 1x1 convolution)
                        long arow = grow;
                        bool aon = true;
                        if (p.add_W > 0) {
                            const int hw = p.add_H * p.add_W, bb = grow / hw, q = grow - bb * hw, h = q / p.add_W, w = q - h * p.add_W;
                            aon = !((h | w) & 1);
                            arow = ((long)bb * ((p.add_H + 1) >> 1) + (h >> 1)) * ((p.add_W + 1) >> 1) + (w >> 1);
                        }
                        if (aon) {
                            const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addp + arow * p.ldadd + gcol);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)ad[e]);
                        }
                    }
                    *reinterpret_cast<bf16x8*>(Cg + (long)grow * p.ldc + gcol) = v;
                    if (st) {
                        float t1 = 0.f, t2 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = (float)v[e];
                            t1 += f; t2 += f * f;
                            if (csplit && gcol + e >= cb) { c1 += f; c2 += f * f; }
                        }
                        const float hi = grow >= rb ? 1.f : 0.f;
                        sA1 += t1; sA2 += t2; sR1 += hi * t1; sR2 += hi * t2;
                        sC1 += c1; sC2 += c2; sB1 += hi * c1; sB2 += hi * c2;
                    }
                }
            }
            if (st) {
                __shared__ float sst[4][8];
                float vals[8] = {sA1, sA2, sR1, sR2, sC1, sC2, sB1, sB2};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k >= 4 && !csplit) break;       // no group boundary in this tile (block-uniform): the last four are zero
                    float x = vals[k];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
                    vals[k] = x;
                }
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) sst[wave][k] = vals[k];
                }
                __syncthreads();
                if (tid < 8) {
                    // tid = (row half << 2) | (column half << 1) | (0: sum, 1: sum of squares)
                    const int k = tid & 1, ch = (tid >> 1) & 1, rh = tid >> 2;
                    float a_[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) a_[q] = sst[0][q * 2 + k] + sst[1][q * 2 + k] + sst[2][q * 2 + k] + sst[3][q * 2 + k];
                    const float A = a_[0], R = a_[1], Cc = a_[2], Bb = a_[3];
                    const float val = rh ? (ch ? Bb : R - Bb) : (ch ? Cc - Bb : A - R - Cc + Bb);
                    const int row0 = rh ? rb : m0, col0 = ch ? cb : n0;
                    if (row0 < p.M && row0 < m0 + 128 && col0 < p.N && col0 < n0 + 128)
                        atomicAdd(p.gn_sums + ((long)(row0 / p.Tlen) * p.gn_G + col0 / p.gn_Cg) * 2 + k, (double)val);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < (NW ? 1 : 2); ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn * 64 + b * 32 + lr;
            const bool cok = full || (col < p.N);
            const int colc = cok ? col : 0;
            const int rbase = m0 + (NW ? wm * 32 : wm * 64 + a * 32) + 4 * lh;
            if (p.splitk > 1) {
                float* dst = p.partial + ((long)z * (p.M - p.row0) - p.row0) * p.N + colc;      // slab rows are relative to row0
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (cok && (full || row < p.M)) dst[(long)row * p.N] = acc[a][b][r];
                }
            } else {
                const float bv = p.bias ? p.bias[colc] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (cok && (full || row < p.M)) {
                        float v = acc[a][b][r] * sc + bv;
                        if (addp) v += to_f32(addp[(long)row * p.ldadd + col]);
                        if (p.out_f32) reinterpret_cast<float*>(p.C)[(long)row * p.ldc + col] = v;
                        else reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = from_f32<T>(v);
                    }
                }
            }
        }
    }
}

typedef __attribute__((address_space(3))) void lds_void;

// =========================================================================================
// NT, LDS-DMA, 128(M) x 256(N) tile, K-step 64, SOFTWARE-PIPELINED ("wide64p", bf16 only).
//   * operand tiles go global -> LDS directly (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write.  An LDS-DMA
//     wave-instruction writes M0-base + lane*16, i.e. 8 consecutive 128-byte rows; bank conflicts are avoided by an XOR
//     swizzle applied on the SOURCE side: LDS slot p (16 B) of row r holds source chunk p ^ ((r>>1)&7) (two 128-B rows
//     share a 256-B bank row, so the 8 chunk slots x 2 row parities of a ds_read_b128 lane group are all distinct).
//   * masked taps / K tails / tile edges: the lane's buffer offset is pointed out of range and the hardware writes zeros
//     into LDS (verified on MI355X: tests/micro/lds_dma_probe.hip).
//   * 3 stages x 48 KiB = 144 KiB of the 160 KiB LDS, one block (4 waves, 128 fp32 accumulators each in AGPRs) per CU:
//     32 MFMAs per wave per barrier, 12 DMA ops per wave per stage; each wave counts only its own DMA ops, so
//     `s_waitcnt vmcnt(12)` = "my part of the next stage landed, the one after may still be in flight" (96 KiB of operand
//     bytes in flight per CU); the barrier after it publishes all four waves' parts.
//   * with one wave per SIMD every exposed LDS latency stalls the MFMA pipe, so fragments are double-buffered:
//       L(k1) M(k0) | L(k2) M(k1) | L(k3) M(k2) | wait lgkm+vmcnt, barrier, refill THIS stage, L(k0 of next tile) | M(k3)
//     i.e. the reads of sub-step k+1 are issued before the MFMAs of k, and the stage hand-off sits in front of the last
//     MFMA group.  The refill targets the stage just read (all waves have retired their reads: lgkmcnt(0) + barrier).
// History (profiles/r01_summary.md): 128x128 register-staged 520 TFLOP/s on 5120^2 k5 -> 128x128 LDS-DMA 620 -> 128x256
// K-step 32 760 -> K-step 64 930 -> software-pipelined 1030.  The intermediate kernels were removed.  Two
// two-waves-per-SIMD forms were tried after the weight-gradient kernel gained 30 % from that recipe and both lost here
// (same box, 5120^2 k5: this kernel 932-942): K-step 32 with two independent blocks per CU 791-820 (64-byte source rows);
// eight waves in two K-groups half a stage out of phase, partial sums joined through LDS, 878.
// =========================================================================================
__global__ __launch_bounds__(256, 1) void gemm_nt_wide64p_kernel(const GemmNT p) {
    typedef bf16_t T;
    constexpr int EPC = 8, BK = 64, ESZ = 2;
    constexpr int TILEA = 128 * 128, TILEW = 256 * 128;
    constexpr int STAGEB = TILEA + TILEW;       // 48 KiB
    constexpr int NS = 3;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGEB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.N + 255) >> 8, tiles_m = (p.M - p.row0 + 127) >> 7;
    const int ntiles = tiles_n * tiles_m;
    const int logical = xcd_remap(blockIdx.x, ntiles * p.splitk);
    const int z = logical / ntiles;
    const int tile = logical - z * ntiles;
    int tm, tn;
    grouped_raster(tile, tiles_m, tiles_n, tm, tn);
    const int m0 = p.row0 + (tm << 7), n0 = tn << 8;
    const int kchunks = (p.K + BK - 1) / BK;
    const int total = p.taps * kchunks;
    const int s_begin = (int)((long)total * z / p.splitk);
    const int s_end = (int)((long)total * (z + 1) / p.splitk);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (int)p.w_bytes, 0x00020000);
    // DMA roles: one op = 8 rows x 128 B.  Wave w: A rows [32w, 32w+32) (4 ops), W rows [64w, 64w+64) (8 ops).
    const int rl = lane >> 3, dp = lane & 7;
    // tile-local rows of op q: base + 8q + rl; (row>>1)&7 = ((8q + rl)>>1)&7 since bases are multiples of 32
    // -> the swizzle term only depends on (8q + rl): ((rl>>1) + 4q) & 7
    const int lda_b = (int)(p.lda * ESZ), ldw_b = (int)(p.ldw * ESZ), wts_b = (int)(p.w_tap_stride * ESZ);
    uint32_t aoffs[4], woffs[8];
    uint32_t amask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 32 + q * 8 + rl;
        const int dc = dp ^ ((row >> 1) & 7);
        const int am = m0 + row;
        aoffs[q] = (uint32_t)((long)am * lda_b + dc * 16);
        uint32_t mk = 0u;
        const int at = am % p.Tlen;
        for (int j = 0; j < p.taps; ++j)
            if (am < p.M && (unsigned)(at + j - p.pad) < (unsigned)p.Tlen) mk |= 1u << j;
        amask[q] = mk | ((uint32_t)dc << 8);      // bits 8..10: source chunk (for the K-tail test)
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = wave * 64 + q * 8 + rl;
        const int dc = dp ^ ((row >> 1) & 7);
        const int wn = n0 + row;
        woffs[q] = wn < p.N ? (uint32_t)((long)wn * ldw_b + dc * 16) : OOB_OFF;
    }
    // all 12 ops of a lane use source chunks dp ^ s with s in 0..7; K-tail validity is per chunk:
    // chunk c of the current step is valid iff ld_kcb + 16*c < K bytes
    unsigned char* const dmaA = smem + wave * 4096;
    unsigned char* const dmaW = smem + TILEA + wave * 8192;
    int ld_j = 0, ld_kcb = 0, ld_aoff = 0, ld_woff = 0;
    {
        const int kci0 = s_begin / p.taps;
        ld_j = s_begin - kci0 * p.taps;
        ld_kcb = kci0 * BK * ESZ;
        ld_aoff = (ld_j - p.pad) * lda_b + ld_kcb;
        ld_woff = ld_j * wts_b + ld_kcb;
    }
    const int kK_b = p.K * ESZ;
#define SGV_W64_A(Q, STAGE)                                                                                   \
    {                                                                                                         \
        const bool ok_ = ((amask[Q] >> ld_j) & 1u) && ((ld_kcb + (int)((amask[Q] >> 8) & 7u) * 16) < kK_b);   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(dmaA + (STAGE) * STAGEB + (Q) * 1024), 16,  \
                                                 ok_ ? aoffs[Q] + (uint32_t)ld_aoff : OOB_OFF, 0, 0, 0);      \
    }
#define SGV_W64_W(Q, STAGE)                                                                                   \
    {                                                                                                         \
        const int dc_ = dp ^ ((((Q) * 8 + rl) >> 1) & 7);                                                     \
        const bool ok_ = (ld_kcb + dc_ * 16) < kK_b;                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(dmaW + (STAGE) * STAGEB + (Q) * 1024), 16,  \
                                                 ok_ ? woffs[Q] + (uint32_t)ld_woff : OOB_OFF, 0, 0, 0);      \
    }
#define SGV_W64_ISSUE(STAGE)                                                                                  \
    {                                                                                                         \
        SGV_W64_A(0, STAGE) SGV_W64_A(1, STAGE) SGV_W64_A(2, STAGE) SGV_W64_A(3, STAGE)                       \
        SGV_W64_W(0, STAGE) SGV_W64_W(1, STAGE) SGV_W64_W(2, STAGE) SGV_W64_W(3, STAGE)                       \
        SGV_W64_W(4, STAGE) SGV_W64_W(5, STAGE) SGV_W64_W(6, STAGE) SGV_W64_W(7, STAGE)                       \
        ++ld_j; ld_aoff += lda_b; ld_woff += wts_b;                                                           \
        if (ld_j == p.taps) { ld_j = 0; ld_kcb += BK * ESZ; ld_aoff = ld_kcb - p.pad * lda_b; ld_woff = ld_kcb; } \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int swz = (lr >> 1) & 7;                 // rows a*32 + lr / wave*64 + b*32 + lr: bits 1..3 come from lr
    const int a_frag_off = lr * 128;
    const int w_frag_off = TILEA + (wave * 64 + lr) * 128;
    bf16x8 fa0, fa1, fa2, fa3, fb0, fb1;      // fragment set A
    bf16x8 ga0, ga1, ga2, ga3, gb0, gb1;      // fragment set B
#define SGV_P_LOAD(X, STAGE, KS)                                                                              \
    {                                                                                                         \
        const unsigned char* sa_ = smem + (STAGE) * STAGEB + a_frag_off;                                      \
        const unsigned char* sw_ = smem + (STAGE) * STAGEB + w_frag_off;                                      \
        const int po_ = (((KS) * 2 + lh) ^ swz) * 16;                                                         \
        X##b0 = *reinterpret_cast<const bf16x8*>(sw_ + po_);                                                  \
        X##b1 = *reinterpret_cast<const bf16x8*>(sw_ + 32 * 128 + po_);                                       \
        X##a0 = *reinterpret_cast<const bf16x8*>(sa_ + po_);                                                  \
        X##a1 = *reinterpret_cast<const bf16x8*>(sa_ + 32 * 128 + po_);                                       \
        X##a2 = *reinterpret_cast<const bf16x8*>(sa_ + 64 * 128 + po_);                                       \
        X##a3 = *reinterpret_cast<const bf16x8*>(sa_ + 96 * 128 + po_);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
#define SGV_P_MMA(X)                                                                                          \
    {                                                                                                         \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##a0, X##b0, acc[0][0], 0, 0, 0);                \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##a0, X##b1, acc[0][1], 0, 0, 0);                \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##a1, X##b0, acc[1][0], 0, 0, 0);                \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##a1, X##b1, acc[1][1], 0, 0, 0);                \
        acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##a2, X##b0, acc[2][0], 0, 0, 0);                \
        acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##a2, X##b1, acc[2][1], 0, 0, 0);                \
        acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##a3, X##b0, acc[3][0], 0, 0, 0);                \
        acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##a3, X##b1, acc[3][1], 0, 0, 0);                \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    // one tile: entered with set f = fragments (tile, k0); HANDOFF is the code between M(k2) and M(k3)
#define SGV_P_TILE(CUR, HANDOFF)                                                                              \
    {                                                                                                         \
        SGV_P_LOAD(g, CUR, 1) SGV_P_MMA(f)                                                                    \
        SGV_P_LOAD(f, CUR, 2) SGV_P_MMA(g)                                                                    \
        SGV_P_LOAD(g, CUR, 3) SGV_P_MMA(f)                                                                    \
        HANDOFF                                                                                               \
        SGV_P_MMA(g)                                                                                          \
    }
    // hand-off with a next tile: retire my reads of this stage, wait for my part of the next tile (12 newer DMA
    // ops may stay in flight), publish, refill this stage (if a tile is left), read the next tile's first fragments
#define SGV_P_HAND(CUR, NXT, WAIT, REFILL)                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(" #WAIT ")\n\ts_barrier" ::: "memory");             \
    if (REFILL) { SGV_W64_ISSUE(CUR); }                                                                       \
    asm volatile("" ::: "memory");                                                                            \
    SGV_P_LOAD(f, NXT, 0)
    const int nst = s_end - s_begin;
    if (nst > 0) {
        SGV_W64_ISSUE(0);
        if (nst > 1) SGV_W64_ISSUE(1);
        if (nst > 2) SGV_W64_ISSUE(2);
        if (nst > 2) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
        else if (nst > 1) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        SGV_P_LOAD(f, 0, 0)
        int i = 0;
        // steady state: tiles i, i+1, i+2 in stages 0,1,2; every hand-off still has tile i+3 to issue and tile i+2
        // in flight behind the one it waits for
        while (i + 6 <= nst) {
            SGV_P_TILE(0, SGV_P_HAND(0, 1, 12, true))
            SGV_P_TILE(1, SGV_P_HAND(1, 2, 12, true))
            SGV_P_TILE(2, SGV_P_HAND(2, 0, 12, true))
            i += 3;
        }
        // 1..5 tiles left, tile i in stage 0; tiles up to min(nst, i+3)-1 are already issued
        int rem = nst - i;
        if (rem == 5) {        // issue i+3, i+4 then drain
            SGV_P_TILE(0, SGV_P_HAND(0, 1, 12, true))
            SGV_P_TILE(1, SGV_P_HAND(1, 2, 12, true))
            SGV_P_TILE(2, SGV_P_HAND(2, 0, 12, false))
            SGV_P_TILE(0, SGV_P_HAND(0, 1, 0, false))
            SGV_P_TILE(1, )
        } else if (rem == 4) {
            SGV_P_TILE(0, SGV_P_HAND(0, 1, 12, true))
            SGV_P_TILE(1, SGV_P_HAND(1, 2, 12, false))
            SGV_P_TILE(2, SGV_P_HAND(2, 0, 0, false))
            SGV_P_TILE(0, )
        } else if (rem == 3) {
            SGV_P_TILE(0, SGV_P_HAND(0, 1, 12, false))
            SGV_P_TILE(1, SGV_P_HAND(1, 2, 0, false))
            SGV_P_TILE(2, )
        } else if (rem == 2) {
            SGV_P_TILE(0, SGV_P_HAND(0, 1, 0, false))
            SGV_P_TILE(1, )
        } else {
            SGV_P_TILE(0, )
        }
    }
#undef SGV_P_LOAD
#undef SGV_P_MMA
#undef SGV_P_TILE
#undef SGV_P_HAND
#undef SGV_W64_A
#undef SGV_W64_W
#undef SGV_W64_ISSUE

    // ---- epilogue: bf16 tiles go through LDS for 16-byte coalesced stores; split-K / fp32 outputs store directly ----
    const float sc = p.scale ? *p.scale : 1.0f;
    const T* addp = reinterpret_cast<const T*>(p.addend);
    if (p.splitk == 1 && !p.out_f32) {
        constexpr int CP = 528;
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int lcol = wave * 64 + b * 32 + lr;
                const int gcol = n0 + lcol;
                const float bv = (p.bias && gcol < p.N) ? p.bias[gcol] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lrow = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    *reinterpret_cast<bf16_t*>(smem + lrow * CP + lcol * 2) = (bf16_t)(acc[a][b][r] * sc + bv);
                }
            }
        }
        __syncthreads();
        bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = tid + i * 256;
            const int lrow = c >> 5, lc8 = (c & 31) * 8;
            const int grow = m0 + lrow, gcol = n0 + lc8;
            if (grow < p.M && gcol < p.N) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + lrow * CP + lc8 * 2);
                if (addp) {
                    const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addp + (long)grow * p.ldadd + gcol);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)ad[e]);
                }
                *reinterpret_cast<bf16x8*>(Cg + (long)grow * p.ldc + gcol) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wave * 64 + b * 32 + lr;
            const bool cok = col < p.N;
            const int colc = cok ? col : 0;
            const int rbase = m0 + a * 32 + 4 * lh;
            if (p.splitk > 1) {
                float* dst = p.partial + ((long)z * (p.M - p.row0) - p.row0) * p.N + colc;      // slab rows are relative to row0
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (cok && row < p.M) dst[(long)row * p.N] = acc[a][b][r];
                }
            } else {
                const float bv = p.bias ? p.bias[colc] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (cok && row < p.M) {
                        float v = acc[a][b][r] * sc + bv;
                        if (addp) v += to_f32(addp[(long)row * p.ldadd + col]);
                        reinterpret_cast<float*>(p.C)[(long)row * p.ldc + col] = v;
                    }
                }
            }
        }
    }
}

// split-K combine: out = scale * sum_z partial[z] + bias + addend
// N % 4 == 0 (every bf16 layer): four columns per thread, 16-byte slab loads, one division per four elements
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_reduce4_kernel(const GemmNT p) {
    const long total = (long)(p.M - p.row0) * p.N, quads = total >> 2;
    const float sc = p.scale ? *p.scale : 1.0f;
    const int nq = p.N >> 2;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long)gridDim.x * 256) {
        const int rrow = (int)(q / nq), col = (int)(q - (long)rrow * nq) * 4;
        const int row = rrow + p.row0;
        float4 v = *reinterpret_cast<const float4*>(p.partial + q * 4);
        for (int z = 1; z < p.splitk; ++z) {
            const float4 w = *reinterpret_cast<const float4*>(p.partial + (long)z * total + q * 4);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        float o[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
        if (p.bias) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
            o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w;
        }
        if (p.addend) {
            const T* ad = reinterpret_cast<const T*>(p.addend) + (long)row * p.ldadd + col;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += to_f32(ad[e]);
        }
        if (p.out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)row * p.ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            T* dst = reinterpret_cast<T*>(p.C) + (long)row * p.ldc + col;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = from_f32<T>(o[e]);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_reduce_kernel(const GemmNT p) {
    const long total = (long)(p.M - p.row0) * p.N;
    const float sc = p.scale ? *p.scale : 1.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int rrow = (int)(i / p.N), col = (int)(i - (long)rrow * p.N);
        const int row = rrow + p.row0;
        float v = 0.f;
        for (int z = 0; z < p.splitk; ++z) v += p.partial[(long)z * total + i];
        v = v * sc + (p.bias ? p.bias[col] : 0.f);
        if (p.addend) v += to_f32(reinterpret_cast<const T*>(p.addend)[(long)row * p.ldadd + col]);
        if (p.out_f32) reinterpret_cast<float*>(p.C)[(long)row * p.ldc + col] = v;
        else reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = from_f32<T>(v);
    }
}

// =========================================================================================
// TN
// =========================================================================================
// C2D: the X operand is a virtual im2col matrix (GemmTN::cv_*): a thread's 16-byte column chunk belongs to one window
// offset (kh, kw), its rows are output pixels whose (oh, ow) is carried from stage to stage
template <typename T, bool USE_TR, bool C2D = false>
__global__ __launch_bounds__(256, SGV_GEMM_MIN_WAVES) void gemm_tn_kernel(const GemmTN p) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr bool IS_BF16 = sizeof(T) == 2;
    constexpr int KR = IS_BF16 ? 32 : 16;      // reduction rows (m) per step
    constexpr int CPR = 128 / EPC;             // 16-byte chunks per tile row
    // bf16: pitch 320 B == 64 (mod 256) so the 4 k-rows of a ds_read_b64_tr_b16 block hit disjoint banks
    constexpr int LD = IS_BF16 ? 160 : 128;
    constexpr int ROWB = LD * (int)sizeof(T);
    constexpr int TILEB = KR * ROWB;
    constexpr int LPT = KR * CPR / 256;
    constexpr int RSTEP = 256 / CPR;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILEB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_2 = (p.N2 + 127) >> 7, tiles_1 = (p.N1 + 127) >> 7;
    const int ntiles = tiles_1 * tiles_2;
    // 1-D grid: XCD-chunked, (tap, split-K slice) slowest, grouped tile raster within.
    const int logical = xcd_remap(blockIdx.x, ntiles * p.taps * p.splitk);
    const int tz = logical / ntiles;
    const int tile = logical - tz * ntiles;
    int t1, t2;
    grouped_raster(tile, tiles_1, tiles_2, t1, t2);
    const int i0 = t1 << 7, j0 = t2 << 7;
    const int tap = tz / p.splitk, z = tz - tap * p.splitk;
    const int dt = C2D ? 0 : tap - p.pad;
    const int ksteps = (p.M + KR - 1) / KR;
    const int s_begin = (int)((long)ksteps * z / p.splitk);
    const int s_end = (int)((long)ksteps * (z + 1) / p.splitk);

    const int cq = tid % CPR;
    const int r0 = tid / CPR;
    static_assert(LPT == 2, "two 16-byte load slots per operand per thread");
    constexpr int ESZ = (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)p.b_bytes, 0x00020000);
    const bool a_cok = (i0 + cq * EPC) < p.N1;
    const bool b_cok = (j0 + cq * EPC) < p.N2;
    const uint32_t acol = a_cok ? (uint32_t)((i0 + cq * EPC) * ESZ) : OOB_OFF;
    const uint32_t lda_b = (uint32_t)(p.lda * ESZ), ldb_b = (uint32_t)(p.ldb * ESZ);
    uint32_t bcol = b_cok ? (uint32_t)((j0 + cq * EPC) * ESZ) : OOB_OFF;
    int dh = 0, dw = 0;                        // C2D: this thread's window offset minus the padding
    if constexpr (C2D) {
        const int vc = j0 + cq * EPC;
        const int tp = vc / p.cv_C, kh = tp / p.cv_kw;
        dh = kh - p.cv_P; dw = tp - kh * p.cv_kw - p.cv_P;
        if (b_cok) bcol = (uint32_t)((vc - tp * p.cv_C) * ESZ);
    }
    uint4 ra0A, ra1A, rb0A, rb1A, ra0B, ra1B, rb0B, rb1B;

    // rows past M and taps that leave the sample window point out of range -> hardware returns zeros
    // incremental row state of the next tile to load: row indices, their time index within the sample
    int ld_m0 = s_begin * KR + r0, ld_m1 = ld_m0 + RSTEP;
    int ld_t0 = ld_m0 % p.Tlen, ld_t1 = ld_m1 % p.Tlen;
    uint32_t ld_a0 = (uint32_t)ld_m0 * lda_b + acol, ld_a1 = (uint32_t)ld_m1 * lda_b + acol;
    uint32_t ld_b0 = (uint32_t)(ld_m0 + dt) * ldb_b + bcol, ld_b1 = (uint32_t)(ld_m1 + dt) * ldb_b + bcol;
    const int kr_t = KR % p.Tlen;
    // C2D: (image byte offset, oh, ow) of the two rows
    uint32_t im0 = 0u, im1 = 0u;
    int oh0 = 0, ow0 = 0, oh1 = 0, ow1 = 0;
    const uint32_t img_b = C2D ? (uint32_t)p.cv_H * (uint32_t)p.cv_W * ldb_b : 0u;
    if constexpr (C2D) {
        const int hw = p.cv_Ho * p.cv_Wo;
        const int b0_ = ld_m0 / hw, q0_ = ld_m0 - b0_ * hw, b1_ = ld_m1 / hw, q1_ = ld_m1 - b1_ * hw;
        oh0 = q0_ / p.cv_Wo; ow0 = q0_ - oh0 * p.cv_Wo; im0 = (uint32_t)b0_ * img_b;
        oh1 = q1_ / p.cv_Wo; ow1 = q1_ - oh1 * p.cv_Wo; im1 = (uint32_t)b1_ * img_b;
    }
#define SGV_TN_ROW2D(PB, LB, PA, IM, OH, OW)                                                                  \
    {                                                                                                         \
        const int ih_ = OH * p.cv_S + dh, iw_ = OW * p.cv_S + dw;                                             \
        PB = PA && (unsigned)ih_ < (unsigned)p.cv_H && (unsigned)iw_ < (unsigned)p.cv_W;                      \
        LB = IM + (uint32_t)(ih_ * p.cv_W + iw_) * ldb_b + bcol;                                              \
        OW += KR;                                                                                             \
        while (OW >= p.cv_Wo) { OW -= p.cv_Wo; if (++OH == p.cv_Ho) { OH = 0; IM += img_b; } }                \
    }
#define SGV_TN_GLOAD(S, X)                                                                                    \
    {                                                                                                         \
        const bool pa0 = (ld_m0 < p.M), pa1 = (ld_m1 < p.M);                                                  \
        bool pb0, pb1;                                                                                        \
        if constexpr (C2D) {                                                                                  \
            SGV_TN_ROW2D(pb0, ld_b0, pa0, im0, oh0, ow0) SGV_TN_ROW2D(pb1, ld_b1, pa1, im1, oh1, ow1)         \
        } else {                                                                                              \
            pb0 = pa0 && ((unsigned)(ld_t0 + dt) < (unsigned)p.Tlen);                                         \
            pb1 = pa1 && ((unsigned)(ld_t1 + dt) < (unsigned)p.Tlen);                                         \
        }                                                                                                     \
        ra0##X = bload16(rsA, pa0 ? ld_a0 : OOB_OFF);                                                         \
        ra1##X = bload16(rsA, pa1 ? ld_a1 : OOB_OFF);                                                         \
        rb0##X = bload16(rsB, pb0 ? ld_b0 : OOB_OFF);                                                         \
        rb1##X = bload16(rsB, pb1 ? ld_b1 : OOB_OFF);                                                         \
        ld_m0 += KR; ld_m1 += KR;                                                                             \
        if constexpr (!C2D) {                                                                                 \
            ld_t0 += kr_t; if (ld_t0 >= p.Tlen) ld_t0 -= p.Tlen;                                              \
            ld_t1 += kr_t; if (ld_t1 >= p.Tlen) ld_t1 -= p.Tlen;                                              \
            ld_b0 += KR * ldb_b; ld_b1 += KR * ldb_b;                                                         \
        }                                                                                                     \
        ld_a0 += KR * lda_b; ld_a1 += KR * lda_b;                                                             \
    }
#define SGV_TN_SSTORE(BUF, X)                                                                                 \
    {                                                                                                         \
        unsigned char* sa_ = smem + (BUF) * 2 * TILEB + r0 * ROWB + cq * 16;                                  \
        unsigned char* sb_ = sa_ + TILEB;                                                                     \
        *reinterpret_cast<uint4*>(sa_) = ra0##X;                                                              \
        *reinterpret_cast<uint4*>(sa_ + RSTEP * ROWB) = ra1##X;                                               \
        *reinterpret_cast<uint4*>(sb_) = rb0##X;                                                              \
        *reinterpret_cast<uint4*>(sb_ + RSTEP * ROWB) = rb1##X;                                               \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#define SGV_TN_COMPUTE(BUF)                                                                                   \
    {                                                                                                         \
        const unsigned char* sa_ = smem + (BUF) * 2 * TILEB;                                                  \
        const unsigned char* sb_ = sa_ + TILEB;                                                               \
        if constexpr (IS_BF16) {                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < KR / 16; ++ks) {                                          \
                bf16x8 a0_, a1_, b0_, b1_;                                                                    \
                if constexpr (USE_TR) {                                                                       \
                    /* ds_read_b64_tr_b16: per 16-lane group g a 4(k) x 16(col) block; lane 4q+p of the  */   \
                    /* group addresses row q, cols 4p..4p+3; lane i receives column i, rows 0..3.        */   \
                    const int g_ = lane >> 4, q_ = (lane >> 2) & 3, pp_ = lane & 3;                           \
                    const int kb_ = ks * 16 + 8 * (g_ >> 1) + q_;                                             \
                    const int ca_ = wm * 64 + 16 * (g_ & 1) + 4 * pp_;                                        \
                    const int cb_ = wn * 64 + 16 * (g_ & 1) + 4 * pp_;                                        \
                    const unsigned char* pa_ = sa_ + (kb_ * LD + ca_) * 2;                                    \
                    const unsigned char* pb_ = sb_ + (kb_ * LD + cb_) * 2;                                    \
                    const s16x4 a0l_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa_));            \
                    const s16x4 a0h_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa_ + 4 * LD * 2));   \
                    const s16x4 a1l_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa_ + 64));       \
                    const s16x4 a1h_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa_ + 64 + 4 * LD * 2)); \
                    const s16x4 b0l_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb_));            \
                    const s16x4 b0h_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb_ + 4 * LD * 2));   \
                    const s16x4 b1l_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb_ + 64));       \
                    const s16x4 b1h_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb_ + 64 + 4 * LD * 2)); \
                    a0_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0l_, a0h_, 0, 1, 2, 3, 4, 5, 6, 7)); \
                    a1_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a1l_, a1h_, 0, 1, 2, 3, 4, 5, 6, 7)); \
                    b0_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0l_, b0h_, 0, 1, 2, 3, 4, 5, 6, 7)); \
                    b1_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b1l_, b1h_, 0, 1, 2, 3, 4, 5, 6, 7)); \
                } else {                                                                                      \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                           \
                        const int k_ = ks * 16 + 8 * lh + e;                                                  \
                        a0_[e] = *reinterpret_cast<const bf16_t*>(sa_ + (k_ * LD + wm * 64 + lr) * 2);        \
                        a1_[e] = *reinterpret_cast<const bf16_t*>(sa_ + (k_ * LD + wm * 64 + 32 + lr) * 2);   \
                        b0_[e] = *reinterpret_cast<const bf16_t*>(sb_ + (k_ * LD + wn * 64 + lr) * 2);        \
                        b1_[e] = *reinterpret_cast<const bf16_t*>(sb_ + (k_ * LD + wn * 64 + 32 + lr) * 2);   \
                    }                                                                                         \
                }                                                                                             \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b0_, acc[0][0], 0, 0, 0);            \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b1_, acc[0][1], 0, 0, 0);            \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b0_, acc[1][0], 0, 0, 0);            \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b1_, acc[1][1], 0, 0, 0);            \
            }                                                                                                 \
        } else {                                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < KR / 2; ++ks) {                                           \
                const int k_ = ks * 2 + lh;                                                                   \
                const float a0_ = *reinterpret_cast<const float*>(sa_ + (k_ * LD + wm * 64 + lr) * 4);        \
                const float a1_ = *reinterpret_cast<const float*>(sa_ + (k_ * LD + wm * 64 + 32 + lr) * 4);   \
                const float b0_ = *reinterpret_cast<const float*>(sb_ + (k_ * LD + wn * 64 + lr) * 4);        \
                const float b1_ = *reinterpret_cast<const float*>(sb_ + (k_ * LD + wn * 64 + 32 + lr) * 4);   \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b0_, acc[0][0], 0, 0, 0);               \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b1_, acc[0][1], 0, 0, 0);               \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b0_, acc[1][0], 0, 0, 0);               \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b1_, acc[1][1], 0, 0, 0);               \
            }                                                                                                 \
        }                                                                                                     \
    }

#define SGV_TN_STEP(XL, XS)                                                                                  \
    {                                                                                                         \
        SGV_TN_GLOAD(s + 2, XL);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        SGV_TN_COMPUTE(cur);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        SGV_TN_SSTORE(cur ^ 1, XS);                                                                           \
        __syncthreads();                                                                                      \
        cur ^= 1; ++s;                                                                                        \
    }
    if (s_begin < s_end) {
        int s = s_begin, cur = 0;
        SGV_TN_GLOAD(s, A);
        if (s + 1 < s_end) SGV_TN_GLOAD(s + 1, B);
        SGV_TN_SSTORE(0, A);
        __syncthreads();
        while (s + 3 < s_end) {
            SGV_TN_STEP(A, B)
            SGV_TN_STEP(B, A)
        }
        const int rem = s_end - s;
        if (rem == 3) {
            SGV_TN_STEP(A, B)
            SGV_TN_COMPUTE(cur);
            SGV_TN_SSTORE(cur ^ 1, A);
            __syncthreads();
            cur ^= 1;
            SGV_TN_COMPUTE(cur);
        } else if (rem == 2) {
            SGV_TN_COMPUTE(cur);
            SGV_TN_SSTORE(cur ^ 1, B);
            __syncthreads();
            cur ^= 1;
            SGV_TN_COMPUTE(cur);
        } else {
            SGV_TN_COMPUTE(cur);
        }
    }
#undef SGV_TN_STEP
#undef SGV_TN_ROW2D
#undef SGV_TN_GLOAD
#undef SGV_TN_SSTORE
#undef SGV_TN_COMPUTE

    // split-K: every (tile, tap, z) block owns its slab region -> plain stores, deterministic
    float* outp = p.out + (long)z * p.out_slab_stride + (long)tap * p.out_tap_stride;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = j0 + wn * 64 + b * 32 + lr;
            if (col >= p.N2) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= p.N1) continue;
                outp[(long)row * p.ldo + col] = acc[a][b][r];
            }
        }
    }
}

// =========================================================================================
// TN, LDS-DMA, 128 (N1) x 256 (N2) tile, 32 reduction rows per stage, TWO blocks per CU ("tn_w2", bf16 only).
// The operands are m-major, so a stage is 32 rows x 256 B of dY and 32 rows x 512 B of X exactly as they lie in memory
// (HBM/L2 -> LDS with buffer_load ... lds, 3 x 24 KiB ring, counted vmcnt), and MFMA fragments come from
// ds_read_b64_tr_b16.  A 16-lane group of that instruction reads 4 consecutive rows x 32 B: with 256/512-byte row
// pitches those rows would share banks, so the 16-byte chunk c of row r is stored at chunk slot c ^ ((r & 3) << 2)
// (the DMA applies the permutation on the source side, the fragment reads on the LDS side): the four rows land in four
// different 64-byte bank groups.  The 64x128 per-wave tile keeps LDS traffic at 0.75 KiB per MFMA (the 128x128 kernel
// needs 1 KiB, which is the LDS peak at full MFMA rate); 72 KiB of LDS and 198 registers let two blocks share a CU, so
// one block's stage hand-off, prologue and epilogue hide under the other's MFMAs.  Wave w DMAs rows [8w, 8w+8) of both
// operands (2 + 4 ops per stage, vmcnt(6)).  Needs Tlen >= 32 (one conditional subtract keeps the per-row time index).
// The transposed reads and the boundary stores are inline asm with hand-tracked lgkmcnt: the builtin / plain C++ LDS
// accesses make the compiler put s_waitcnt vmcnt(0) in front of them while LDS-DMA is in flight.
// =========================================================================================
// C2D: X is the virtual im2col operand of GemmTN::cv_* -- a lane's chunk column fixes its window offset (kh, kw), the rows'
// (image, oh, ow) are carried from stage to stage and every DMA is predicated on its pixel lying inside the image.
template <bool C2D>
__global__ __launch_bounds__(256, 2) void gemm_tn_w2_kernel(const GemmTN p) {
    constexpr int ESZ = 2, KR = 32;
    constexpr int ROWA = 256, ROWX = 512;
    constexpr int TILEA = KR * ROWA, TILEX = KR * ROWX;
    constexpr int STAGEB = TILEA + TILEX;       // 24 KiB
    constexpr int NS = 3;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGEB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_1 = (p.N1 + 127) >> 7, tiles_2 = (p.N2 + 255) >> 8;
    const int ntiles = tiles_1 * tiles_2;
    const int nitems = ntiles * p.taps * p.splitk;
    // ---- schedule.  order 0: one work item per block, XCD-chunked, (tap, slice) slowest, grouped raster (round 1).
    // order 1 (persistent): the grid is 2 blocks per CU; XCD x owns the x-th contiguous chunk of the item list and its block j
    // walks items chunk_lo + j, + nbx, ...: all blocks start together and do equal work per item, so the ~64 items an XCD has in
    // flight stay in step and form one patch of the list.  The list is cut into patches of pt1 x pt2 tiles x all taps (tap
    // fastest: the taps of a tile read the same dY panel and the same X panel, shifted by a row), so the blocks in flight share
    // pt1 dY panels and pt2 X panels in the XCD's L2.
    int item, item_end, item_step;
    if (p.order == 1) {
        const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
        const int q8 = nitems >> 3, r8 = nitems & 7;
        const int it_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        item_end = it_lo + (xcd < r8 ? q8 + 1 : q8);
        item_step = ((int)gridDim.x - xcd + 7) >> 3;
        item = it_lo + jb;
    } else {
        item = xcd_remap(blockIdx.x, nitems);
        item_end = item + 1; item_step = 1;
    }
    const int ksteps = (p.M + KR - 1) / KR;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)p.b_bytes, 0x00020000);
    const int lda_b = (int)(p.lda * ESZ), ldx_b = (int)(p.ldb * ESZ);
    // DMA lane roles.  dY op q: rows 8w + 4q + (lane>>4), slot lane&15.  X op q: rows 8w + 2q + (lane>>5), slot lane&31.
    // No per-row predicates at issue time:
    //  * rows >= M and rows m+dt outside [0, M) fall outside the buffer extents -> the hardware returns zeros;
    //  * columns past N1/N2 start from OOB_OFF (sum stays >= the extent, < 2^32 because row offsets are < 2^31);
    //  * X rows whose tap leaves the sample window (at most |dt| <= 2 rows per boundary) are zeroed in LDS after they
    //    landed (SGV_T2_FIX), only in the stages that contain a sample boundary.
    const int la_row = lane >> 4;
    const int la_chunk = (lane & 15) ^ (la_row << 2);
    const int lx_row = lane >> 5;
    const int lx_chunk0 = (lane & 31) ^ (lx_row << 2);
    const int lx_chunk1 = (lane & 31) ^ ((2 + lx_row) << 2);
    const uint32_t img_b = C2D ? (uint32_t)p.cv_H * (uint32_t)p.cv_W * (uint32_t)ldx_b : 0u;
    unsigned char* const dmaA = smem + wave * 8 * ROWA;
    unsigned char* const dmaX = smem + TILEA + wave * 8 * ROWX;
    const uint32_t zero_base = (uint32_t)(uintptr_t)(lds_void*)smem + TILEA + (tid & 31) * 16;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};

    const int lr = lane & 31, lh = lane >> 5;
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int rowpart = 8 * (g >> 1) + qq;
    const uint32_t smem_b = (uint32_t)(uintptr_t)(lds_void*)smem;
    const uint32_t a_base = smem_b + rowpart * ROWA + 32 * (g & 1) + 8 * pp;
    const uint32_t fa0 = a_base + ((0 ^ qq) << 6), fa1 = a_base + ((1 ^ qq) << 6);
    const uint32_t fa2 = a_base + ((2 ^ qq) << 6), fa3 = a_base + ((3 ^ qq) << 6);
    const int xc0 = (8 * wave + 0 + 2 * (g & 1) + (pp >> 1)) ^ (qq << 2);
    const int xc1 = (8 * wave + 4 + 2 * (g & 1) + (pp >> 1)) ^ (qq << 2);
    const uint32_t fx0 = smem_b + TILEA + rowpart * ROWX + xc0 * 16 + 8 * (pp & 1);
    const uint32_t fx1 = smem_b + TILEA + rowpart * ROWX + xc1 * 16 + 8 * (pp & 1);
    typedef long tr64_t;
    typedef long tr64x2_t __attribute__((ext_vector_type(2)));

#define SGV_T2_A(Q, STAGE)                                                                                    \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(dmaA + (STAGE) + (Q) * 1024), 16,      \
                                             aoffs[Q] + ld_a, 0, 0, 0);
#define SGV_T2_X(Q, STAGE)                                                                                    \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(dmaX + (STAGE) + (Q) * 1024), 16,      \
                                             xoffs[Q] + ld_x, 0, 0, 0);
    // C2D op Q: row (lane's first row + 2Q) of the stage; offset of its pixel or out of range
#define SGV_T2_X2D(Q, STAGE)                                                                                  \
    {                                                                                                         \
        const int ih_ = oh_ * p.cv_S + (((Q) & 1) ? dh1 : dh0), iw_ = ow_ * p.cv_S + (((Q) & 1) ? dw1 : dw0);  \
        const bool ok_ = (unsigned)ih_ < (unsigned)p.cv_H && (unsigned)iw_ < (unsigned)p.cv_W;                \
        const uint32_t vo_ = ok_ ? im_ + (uint32_t)(ih_ * p.cv_W + iw_) * (uint32_t)ldx_b + (((Q) & 1) ? x_col1 : x_col0) : OOB_OFF; \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(dmaX + (STAGE) + (Q) * 1024), 16, vo_, 0, 0, 0); \
        ow_ += 2;                                                                                             \
        while (ow_ >= p.cv_Wo) { ow_ -= p.cv_Wo; if (++oh_ == p.cv_Ho) { oh_ = 0; im_ += img_b; } }           \
    }
#define SGV_T2_ISSUE(STAGE)                                                                                   \
    {                                                                                                         \
        SGV_T2_A(0, STAGE) SGV_T2_A(1, STAGE)                                                                 \
        if constexpr (C2D) {                                                                                  \
            int oh_ = x_oh, ow_ = x_ow; uint32_t im_ = x_im;                                                  \
            SGV_T2_X2D(0, STAGE) SGV_T2_X2D(1, STAGE) SGV_T2_X2D(2, STAGE) SGV_T2_X2D(3, STAGE)               \
            x_ow += KR;                                                                                       \
            while (x_ow >= p.cv_Wo) { x_ow -= p.cv_Wo; if (++x_oh == p.cv_Ho) { x_oh = 0; x_im += img_b; } }  \
        } else {                                                                                              \
            SGV_T2_X(0, STAGE) SGV_T2_X(1, STAGE) SGV_T2_X(2, STAGE) SGV_T2_X(3, STAGE)                       \
        }                                                                                                     \
        ld_a += (uint32_t)(KR * lda_b); ld_x += (uint32_t)(KR * ldx_b);                                       \
    }
    // X rows of stage STAGE (just landed, barrier passed) whose tap leaves the sample window are zeroed in LDS.  Bad rows
    // sit within |dt| <= 2 rows of a sample boundary; candidates: rows 0,1 (run continuing from the previous stage), the
    // four rows around a boundary inside the window, rows 30,31 (run leading into a boundary at row 32/33).  Each of the
    // 8 x 32 threads tests one candidate row and clears one 16-byte chunk of it.  Skipped (no extra barrier) when no
    // boundary is near the window.
#define SGV_T2_FIX(STAGE)                                                                                     \
    if constexpr (!C2D) {                                                                                     \
        const bool near_ = dt != 0 && (rd_t < 2 || rd_t + KR + 2 > p.Tlen);                                   \
        if (near_) {                                                                                          \
            const int rb_ = rd_t == 0 ? 0 : p.Tlen - rd_t;                                                    \
            const int c_ = tid >> 5;                                                                          \
            const int r_ = c_ < 2 ? c_ : (c_ < 6 ? rb_ - 4 + c_ : 24 + c_);                                   \
            int t_ = rd_t + r_;                                                                               \
            if (t_ >= p.Tlen) t_ -= p.Tlen;                                                                   \
            if (r_ >= 0 && r_ < KR && (unsigned)(t_ + dt) >= (unsigned)p.Tlen) {                              \
                const uint32_t za_ = zero_base + (uint32_t)((STAGE) + r_ * ROWX);                    \
                asm volatile("ds_write_b128 %0, %1" :: "v"(za_), "v"(zero4) : "memory");                      \
            }                                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                   \
        }                                                                                                     \
        rd_t += KR; if (rd_t >= p.Tlen) rd_t -= p.Tlen;                                                       \
    }
#define SGV_T2_TR(DST, ADDR, IMM, PITCH)                                                                      \
    {                                                                                                         \
        tr64_t lo_, hi_;                                                                                      \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"           \
                     : "=&v"(lo_), "=&v"(hi_) : "v"(ADDR), "n"(IMM), "n"((IMM) + 4 * (PITCH)));               \
        tr64x2_t pr_; pr_[0] = lo_; pr_[1] = hi_;                                                             \
        DST = __builtin_bit_cast(bf16x8, pr_);                                                                \
    }
    // the stage is a run-time (scalar) byte offset, so the main loop is ONE body: a single live copy of the accumulators
    // (stage-specialised bodies with several loop exits made the allocator hold two, which does not fit 256 registers)
#define SGV_T2_LOAD(X, KS)                                                                                    \
    {                                                                                                         \
        SGV_T2_TR(X##B0, ax0, (KS) * 16 * ROWX, ROWX)                                                         \
        SGV_T2_TR(X##B1, ax1, (KS) * 16 * ROWX, ROWX)                                                         \
        SGV_T2_TR(X##A0, aa0, (KS) * 16 * ROWA, ROWA)                                                         \
        SGV_T2_TR(X##A1, aa1, (KS) * 16 * ROWA, ROWA)                                                         \
        SGV_T2_TR(X##A2, aa2, (KS) * 16 * ROWA, ROWA)                                                         \
        SGV_T2_TR(X##A3, aa3, (KS) * 16 * ROWA, ROWA)                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
#define SGV_T2_WAIT(X, CNT)                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                \
                 : "+v"(X##A0), "+v"(X##A1), "+v"(X##A2), "+v"(X##A3), "+v"(X##B0), "+v"(X##B1));
#define SGV_T2_MMA(X)                                                                                         \
    {                                                                                                         \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##A0, X##B0, acc[0][0], 0, 0, 0);                \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##A0, X##B1, acc[0][1], 0, 0, 0);                \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##A1, X##B0, acc[1][0], 0, 0, 0);                \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##A1, X##B1, acc[1][1], 0, 0, 0);                \
        acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##A2, X##B0, acc[2][0], 0, 0, 0);                \
        acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##A2, X##B1, acc[2][1], 0, 0, 0);                \
        acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##A3, X##B0, acc[3][0], 0, 0, 0);                \
        acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X##A3, X##B1, acc[3][1], 0, 0, 0);                \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }

    for (; item < item_end; item += item_step) {
        // ---- decode the work item ----
        int t1, t2, tap, z;
        if (p.order == 1) {
            const int per_z = ntiles * p.taps;
            z = item / per_z;
            const int r0_ = item - z * per_z;
            const int band_full = p.pt1 * tiles_2 * p.taps;
            const int band = r0_ / band_full, r1_ = r0_ - band * band_full;
            const int first1 = band * p.pt1, g1 = min(p.pt1, tiles_1 - first1);
            const int grp_full = g1 * p.pt2 * p.taps;
            const int grp = r1_ / grp_full, r2_ = r1_ - grp * grp_full;
            const int r3_ = r2_ / p.taps;
            tap = r2_ - r3_ * p.taps;
            const int c_ = r3_ / g1;
            t1 = first1 + (r3_ - c_ * g1);
            t2 = grp * p.pt2 + c_;
        } else {
            const int tz = item / ntiles;
            grouped_raster(item - tz * ntiles, tiles_1, tiles_2, t1, t2);
            tap = tz / p.splitk; z = tz - tap * p.splitk;
        }
        t1 = __builtin_amdgcn_readfirstlane(t1); t2 = __builtin_amdgcn_readfirstlane(t2);
        tap = __builtin_amdgcn_readfirstlane(tap); z = __builtin_amdgcn_readfirstlane(z);
        const int i0 = t1 << 7, j0 = t2 << 8;
        const int dt = C2D ? 0 : tap - p.pad;
        const int s_begin = (int)((long)ksteps * z / p.splitk);
        const int s_end = (int)((long)ksteps * (z + 1) / p.splitk);

        const uint32_t a_col = (i0 + la_chunk * 8) < p.N1 ? (uint32_t)((i0 + la_chunk * 8) * ESZ) : OOB_OFF;
        uint32_t x_col0 = (j0 + lx_chunk0 * 8) < p.N2 ? (uint32_t)((j0 + lx_chunk0 * 8) * ESZ) : OOB_OFF;
        uint32_t x_col1 = (j0 + lx_chunk1 * 8) < p.N2 ? (uint32_t)((j0 + lx_chunk1 * 8) * ESZ) : OOB_OFF;
        // C2D: window offsets (minus the padding) of the lane's two chunk columns; state of the lane's first row of the next stage
        int dh0 = 0, dw0 = 0, dh1 = 0, dw1 = 0;
        uint32_t x_im = 0u;
        int x_oh = 0, x_ow = 0;
        if constexpr (C2D) {
            const int vc0 = j0 + lx_chunk0 * 8, vc1 = j0 + lx_chunk1 * 8;
            const int t0_ = vc0 / p.cv_C, t1_ = vc1 / p.cv_C;
            const int kh0_ = t0_ / p.cv_kw, kh1_ = t1_ / p.cv_kw;
            dh0 = kh0_ - p.cv_P; dw0 = t0_ - kh0_ * p.cv_kw - p.cv_P;
            dh1 = kh1_ - p.cv_P; dw1 = t1_ - kh1_ * p.cv_kw - p.cv_P;
            if (vc0 < p.N2) x_col0 = (uint32_t)((vc0 - t0_ * p.cv_C) * ESZ);
            if (vc1 < p.N2) x_col1 = (uint32_t)((vc1 - t1_ * p.cv_C) * ESZ);
            const int m_ = s_begin * KR + wave * 8 + lx_row;
            const int hw_ = p.cv_Ho * p.cv_Wo;
            const int b_ = m_ / hw_, q_ = m_ - b_ * hw_;
            x_oh = q_ / p.cv_Wo; x_ow = q_ - x_oh * p.cv_Wo; x_im = (uint32_t)b_ * img_b;
        }
        uint32_t aoffs[2], xoffs[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) aoffs[q] = (uint32_t)(wave * 8 + q * 4 + la_row) * (uint32_t)lda_b + a_col;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            xoffs[q] = (uint32_t)(wave * 8 + q * 2 + lx_row + dt) * (uint32_t)ldx_b + ((q & 1) ? x_col1 : x_col0);
        uint32_t ld_a = (uint32_t)(s_begin * KR) * (uint32_t)lda_b;
        uint32_t ld_x = (uint32_t)(s_begin * KR) * (uint32_t)ldx_b;
        int rd_t = (s_begin * KR) % p.Tlen;

        f32x16 acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        bf16x8 fA0, fA1, fA2, fA3, fB0, fB1;
        bf16x8 gA0, gA1, gA2, gA3, gB0, gB1;

        const int nst = s_end - s_begin;
        if (nst > 0) {
            SGV_T2_ISSUE(0);
            if (nst > 1) SGV_T2_ISSUE(STAGEB);
            if (nst > 2) SGV_T2_ISSUE(2 * STAGEB);
            if (nst > 2) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
            else if (nst > 1) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            SGV_T2_FIX(0)
            int cur = 0;                                // byte offset of the stage being consumed
            for (int i = 0; i < nst; ++i) {
                const uint32_t aa0 = fa0 + cur, aa1 = fa1 + cur, aa2 = fa2 + cur, aa3 = fa3 + cur;
                const uint32_t ax0 = fx0 + cur, ax1 = fx1 + cur;
                // both sub-steps' fragments are requested up front; the other block on the CU covers the latency
                SGV_T2_LOAD(f, 0) SGV_T2_LOAD(g, 1)
                SGV_T2_WAIT(f, 12) SGV_T2_MMA(f)
                SGV_T2_WAIT(g, 0) SGV_T2_MMA(g)
                if (i + 1 < nst) {
                    // my reads of `cur` have retired (lgkmcnt(0) above): wait for my part of the next stage, publish, refill
                    // `cur` with stage i + 3, patch the next stage's tap-boundary rows
                    const int nxt = cur == 2 * STAGEB ? 0 : cur + STAGEB;
                    if (i + 3 < nst) {
                        asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
                        SGV_T2_ISSUE(cur);
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                    }
                    asm volatile("" ::: "memory");
                    SGV_T2_FIX(nxt)
                    cur = nxt;
                }
            }
        }
        // every wave's LDS reads of this item have retired (lgkmcnt(0) before its last MFMAs): after this barrier the next item's
        // first stages may overwrite the ring
        if (item + item_step < item_end) asm volatile("s_barrier" ::: "memory");

        float* outp = p.out + (long)z * p.out_slab_stride + (long)tap * p.out_tap_stride;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = j0 + wave * 64 + b * 32 + lr;
                if (col >= p.N2) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row >= p.N1) continue;
                    outp[(long)row * p.ldo + col] = acc[a][b][r];
                }
            }
        }
    }
#undef SGV_T2_A
#undef SGV_T2_X
#undef SGV_T2_X2D
#undef SGV_T2_ISSUE
#undef SGV_T2_FIX
#undef SGV_T2_TR
#undef SGV_T2_LOAD
#undef SGV_T2_WAIT
#undef SGV_T2_MMA
}

// =========================================================================================
// host launchers
// =========================================================================================
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Split-K choice: minimise (occupancy rounds) x (K steps per block) + the slab combine pass.
// 256 CUs x 3 resident 256-thread blocks (40 KB LDS, ~150 VGPR) = 768 slots; a grid of 800 blocks takes
// two rounds, so e.g. 200 tiles prefer 3 slices (600 blocks) over 4.
// The 128x256 LDS-DMA kernel is used for bf16 GEMMs with N >= 256 and >= 64 K-steps (short-K shapes are
// faster on the 128x128 kernel: 3 blocks/CU hide the prologue/epilogue).  fp32 (validation mode) stays on
// the 128x128 kernels: the fp32 build of the wide kernel showed an intermittent, unexplained corruption of a
// few accumulator rows with 5-tap convs (tests/micro/nt_sweep.py), while the bf16 build is clean under
// tests/test_kernels_gpu.py::test_gemm_nt_wide_stress.
static bool gemm_nt_is_wide(int dtype, int N, long steps) {
    static const int use_wide = getenv("SGV_GEMM_WIDE") ? atoi(getenv("SGV_GEMM_WIDE")) : 1;
    static const int min_steps = getenv("SGV_WIDE_MIN_STEPS") ? atoi(getenv("SGV_WIDE_MIN_STEPS")) : 64;
    return use_wide && dtype == 1 && N >= 256 && steps >= min_steps;
}
static int pick_splitk(long tiles, long steps, double slab_bytes_per_slice, int min_steps, double slots = 768.0) {
    int best = 1;
    double best_cost = 1e30;
    for (int sk = 1; sk <= 32; ++sk) {
        if (sk > 1 && steps / sk < min_steps) break;
        const double rounds = ceil((double)tiles * sk / slots);
        const double per = ceil((double)steps / sk) + 8.0;             // + prologue/epilogue per block
        double cost = rounds * per;
        if (sk > 1) cost += (2.0 * sk * slab_bytes_per_slice / 3.0e12) / 0.6e-6;   // combine pass, in step units
        if (cost < best_cost * 0.97) { best_cost = cost; best = sk; }
    }
    return best;
}

bool gemm_nt_can_fuse_stats(int dtype, int M, int N, int K, int taps, int Tlen, int Cg) {
    static const int on = getenv("SGV_GEMM_STATS") ? atoi(getenv("SGV_GEMM_STATS")) : 1;
    return on && dtype == 1 && Tlen >= 128 && Cg >= 128 && !gemm_nt_is_wide(dtype, N, (long)taps * cdiv(K, 32)) &&
           gemm_nt_pick_splitk(M, N, K, taps, dtype) == 1;
}
bool gemm_nt_uses_wide(int dtype, int N, int K, int taps) {
    return gemm_nt_is_wide(dtype, N, (long)taps * cdiv(K, dtype == 1 ? 32 : 16));
}
int gemm_nt_pick_splitk(int M, int N, int K, int taps, int dtype) {
    const int bk = dtype == 1 ? 32 : 16;
    const long total = (long)taps * cdiv(K, bk);
    if (gemm_nt_is_wide(dtype, N, total))   // 128x256 tiles, 1 block (4 waves, 312 registers) per CU
        return pick_splitk((long)cdiv(M, 128) * cdiv(N, 256), total, (double)M * N * 4.0, 16, 256.0);
    const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
    return pick_splitk(tiles, total, (double)M * N * 4.0, 8);
}

// 128x256 tiles, two blocks per CU (gemm_tn_w2_kernel): the default for bf16 weight gradients with N2 >= 256; SGV_TN_W2=0
// falls back to the 128x128 register-staged kernel (A/B runs), GemmTN::force_w2 selects it regardless of the env.
static bool gemm_tn_w2_eligible(int dtype, int M, int N1, int N2, int Tlen) {
    return dtype == 1 && N2 >= 256 && N1 >= 64 && Tlen >= 32 && M >= 256;
}
bool gemm_tn_uses_w2(int dtype, int M, int N1, int N2, int Tlen) {
    static const int use_w2 = getenv("SGV_TN_W2") ? atoi(getenv("SGV_TN_W2")) : 1;
    return use_w2 && gemm_tn_w2_eligible(dtype, M, N1, N2, Tlen);
}
int gemm_tn_pick_splitk(int M, int N1, int N2, int taps, int dtype, int Tlen) {
    {   // the 256 x 256 kernel takes the product whole (its items fill the chip without slices)
        GemmTN q; memset(&q, 0, sizeof(q));
        q.M = M; q.N1 = N1; q.N2 = N2; q.taps = taps; q.Tlen = Tlen; q.lda = N1; q.ldb = N2; q.ldo = N2;
        if (Tlen > 0 && gemm_tn_uses_t256(dtype, q)) return 1;
    }
    if (gemm_tn_uses_w2(dtype, M, N1, N2, Tlen))     // 128x256 tiles, two blocks per CU, 32-row stages
        return pick_splitk((long)cdiv(N1, 128) * cdiv(N2, 256) * taps, cdiv(M, 32), (double)taps * N1 * N2 * 4.0, 12, 512.0);
    const int kr = dtype == 1 ? 32 : 16;
    const long tiles = (long)cdiv(N1, 128) * cdiv(N2, 128) * taps;
    const long total = cdiv(M, kr);
    return pick_splitk(tiles, total, (double)taps * N1 * N2 * 4.0, 4);
}

// profiling hook: the next launch_gemm_nt records this event right after its main kernel (before a split-K combine)
static thread_local hipEvent_t g_main_done = nullptr;
void gemm_nt_main_done_event(hipEvent_t ev) { g_main_done = ev; }

int launch_gemm_nt(int dtype, const GemmNT& p, hipStream_t s) {
    const hipEvent_t main_done = g_main_done;
    g_main_done = nullptr;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return 0;
    const int epc = dtype == 1 ? 8 : 4;
    if (p.K % epc || p.lda % epc || p.ldw % epc || p.w_tap_stride % epc) return -1;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return -1;
    if (p.splitk > 1 && !p.partial) return -1;
    const int esz = dtype == 1 ? 2 : 4;
    GemmNT q = p;
    const long arows = p.a_rows > p.M ? p.a_rows : p.M;
    q.a_bytes = ((arows - 1) * p.lda + p.K) * esz;
    q.w_bytes = ((long)(p.taps - 1) * p.w_tap_stride + (long)(p.N - 1) * p.ldw + p.K) * esz;
    if (q.a_bytes >= 0x7FFFFFF0L || q.w_bytes >= 0x7FFFFFF0L) return -1;   // 32-bit buffer offsets
    if (p.row0 < 0 || p.row0 >= p.M || (p.row0 && p.gn_sums)) return -1;
    const int Mr = p.M - p.row0;                                           // rows this launch computes
    const bool c2d = p.cv_kw > 0;
    // strided addend (add_W > 0): the staged bf16 epilogue of gemm_nt_kernel only
    if (p.add_W > 0 && (!p.addend || dtype != 1 || p.splitk != 1 || p.out_f32 || p.add_H < 1 || p.M % (p.add_H * p.add_W) ||
                        (!c2d && gemm_nt_is_wide(dtype, p.N, (long)p.taps * cdiv(p.K, 32)))))
        return -1;
    if (c2d) {
        if (p.taps > 31 || p.taps % p.cv_kw || p.cv_S < 1 || p.cv_P < 0 || p.cv_H < 1 || p.cv_W < 1 || p.cv_Ho < 1 || p.cv_Wo < 1) return -1;
        if (p.M % (p.cv_Ho * p.cv_Wo) || p.a_rows != (long)(p.M / (p.cv_Ho * p.cv_Wo)) * p.cv_H * p.cv_W) return -1;
        // every output pixel's window may start outside the image but must not reach past it by more than the padding
        if ((p.cv_Ho - 1) * p.cv_S - p.cv_P >= p.cv_H || (p.cv_Wo - 1) * p.cv_S - p.cv_P >= p.cv_W) return -1;
        if (p.gn_sums || p.row0) return -1;
    }
    const long total_steps = (long)p.taps * cdiv(p.K, dtype == 1 ? 32 : 16);
    if (p.gn_sums && (dtype != 1 || gemm_nt_is_wide(dtype, p.N, total_steps) || p.splitk != 1 || p.out_f32 || p.Tlen < 128 ||
                      p.gn_Cg < 128 || p.gn_G < 1))
        return -1;                                   // only the bf16 128x128 epilogue accumulates statistics
    // N <= 64 (bf16, no statistics epilogue): the 128 x 64 form of the kernel
    static const int narrow_on = getenv("SGV_GEMM_NARROW") ? atoi(getenv("SGV_GEMM_NARROW")) : 1;
    const bool narrow = narrow_on && dtype == 1 && p.N <= 64 && !p.gn_sums;
    if (narrow) {
        dim3 grid(cdiv(Mr, 128) * p.splitk);
        if (c2d) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, 4, true, true>), grid, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, 4, false, true>), grid, dim3(256), 0, s, q);
    } else if (c2d) {
        dim3 grid(cdiv(Mr, 128) * cdiv(p.N, 128) * p.splitk);
        if (dtype == 1) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, 4, true>), grid, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((gemm_nt_kernel<float, 4, true>), grid, dim3(256), 0, s, q);
    } else if (gemm_nt_is_wide(dtype, p.N, total_steps)) {
        dim3 grid(cdiv(Mr, 128) * cdiv(p.N, 256) * p.splitk);
        hipLaunchKernelGGL(gemm_nt_wide64p_kernel, grid, dim3(256), 0, s, q);
    } else {
        dim3 grid(cdiv(Mr, 128) * cdiv(p.N, 128) * p.splitk);
        if (dtype == 1) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, 4>), grid, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((gemm_nt_kernel<float, 4>), grid, dim3(256), 0, s, q);
    }
    if (main_done) hipEventRecord(main_done, s);
    if (p.splitk > 1) {
        long total = (long)Mr * p.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        const bool quad = p.N % 4 == 0 && p.ldc % 4 == 0 && (!p.addend || p.ldadd % 4 == 0) && (((uintptr_t)p.C) & 15) == 0 &&
                          (!p.bias || (((uintptr_t)p.bias) & 15) == 0);
        if (quad) {
            int b4 = (int)((total / 4 + 255) / 256);
            if (b4 > 4096) b4 = 4096;
            if (dtype == 1) hipLaunchKernelGGL((gemm_nt_reduce4_kernel<bf16_t>), dim3(b4), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((gemm_nt_reduce4_kernel<float>), dim3(b4), dim3(256), 0, s, p);
        } else if (dtype == 1) hipLaunchKernelGGL((gemm_nt_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_nt_reduce_kernel<float>), dim3(blocks), dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Schedule of gemm_tn_w2_kernel (returns the grid).  Persistent form: 512 blocks = two per CU walk an XCD-chunked item list cut
// into patches of pt1 x pt2 tiles x all taps.  The ~64 items an XCD has in flight then read pt1 dY panels (256-byte rows) and
// pt2 X panels (512-byte rows): pick the patch that minimises (pt1 + 2 pt2) / (pt1 pt2) with pt1 pt2 taps <= 64.
// Measured (round 3, tests/micro/order_ab.sh, FETCH_SIZE per launch / time): 5120 x 5120 x 5 taps 2.04 -> 1.15 GB, 829 -> 814 us;
// 2560 x 2560 x 5 taps 0.44 -> 0.19 GB, 212 -> 217 us; the one-tap 95 008-wide gradients do NOT gain (recon head 2.52 -> 2.79 GB,
// 652 -> 719 us; first encoder layer 1.33 -> 1.30 GB, 657 -> 668 us): there the fetches are the 6.5 MB operand falling out of
// the 4 MiB L2 (Infinity-Cache hits), not missed panel sharing.  So: persistent for multi-tap launches with at least four rounds
// of items (SGV_TN_PERSIST=0 / 1 forces one item per block / the persistent walk everywhere).
static int tn_w2_schedule(GemmTN& q, int tiles_1, int tiles_2, int taps) {
    static const int persist_env = getenv("SGV_TN_PERSIST") ? atoi(getenv("SGV_TN_PERSIST")) : -1;
    const bool persist = q.force_w2 == 2 || (persist_env >= 0 ? persist_env != 0 : (taps > 1 && (long)tiles_1 * tiles_2 * taps * q.splitk >= 2048));
    static const int f1 = getenv("SGV_TN_PT1") ? atoi(getenv("SGV_TN_PT1")) : 0;
    static const int f2 = getenv("SGV_TN_PT2") ? atoi(getenv("SGV_TN_PT2")) : 0;
    static const int slots = getenv("SGV_TN_SLOTS") ? atoi(getenv("SGV_TN_SLOTS")) : 512;
    const int nitems = tiles_1 * tiles_2 * taps * q.splitk;
    q.order = persist ? 1 : 0; q.pt1 = 1; q.pt2 = 1;
    if (!persist) return nitems;
    const int cap = slots / 8;
    double best = 1e30;
    for (int a = 1; a <= tiles_1 && a <= cap; ++a)
        for (int b = 1; b <= tiles_2 && a * b * taps <= cap; ++b) {
            const double c = (double)(a + 2 * b) / ((double)a * b);
            if (c < best - 1e-9) { best = c; q.pt1 = a; q.pt2 = b; }
        }
    if (f1 > 0) q.pt1 = f1 < tiles_1 ? f1 : tiles_1;
    if (f2 > 0) q.pt2 = f2 < tiles_2 ? f2 : tiles_2;
    return nitems < slots ? nitems : slots;
}

int launch_gemm_tn(int dtype, const GemmTN& p, hipStream_t s) {
    if (p.M <= 0 || p.N1 <= 0 || p.N2 <= 0) return 0;
    const int epc = dtype == 1 ? 8 : 4;
    if (p.N1 % epc || p.N2 % epc || p.lda % epc || p.ldb % epc) return -1;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return -1;
    if (p.splitk > 1 && p.out_slab_stride <= 0) return -1;
    const int esz = dtype == 1 ? 2 : 4;
    GemmTN q = p;
    q.a_bytes = ((long)(p.M - 1) * p.lda + p.N1) * esz;
    q.b_bytes = ((long)(p.M - 1) * p.ldb + p.N2) * esz;
    const bool c2d = p.cv_kw > 0;
    if (c2d) {
        if (p.taps != 1 || p.cv_C < epc || p.cv_C % epc || p.N2 % p.cv_C || (p.N2 / p.cv_C) % p.cv_kw) return -1;
        if (p.cv_S < 1 || p.cv_P < 0 || p.cv_H < 1 || p.cv_W < 1 || p.cv_Ho < 1 || p.cv_Wo < 1 || p.M % (p.cv_Ho * p.cv_Wo)) return -1;
        q.b_bytes = (((long)(p.M / (p.cv_Ho * p.cv_Wo)) * p.cv_H * p.cv_W - 1) * p.ldb + p.cv_C) * esz;
    }
    if (q.a_bytes >= 0x7FFFFFF0L || q.b_bytes >= 0x7FFFFFF0L) return -1;
    if (!c2d && dtype == 1 && p.use_tr && p.force_w2 >= 0 && p.force_w2 != 1 && p.force_w2 != 2 &&
        (p.force_w2 == 3 ? gemm_tn256_eligible(dtype, p) : gemm_tn_uses_t256(dtype, p))) {
        const int r = launch_gemm_tn256(p, s);
        if (r != -1) return r;                       // -1: shape refused after all (offset ranges): the kernels below take it
    }
    if (c2d && p.use_tr && (gemm_tn_uses_w2(dtype, p.M, p.N1, p.N2, p.M) || (p.force_w2 && gemm_tn_w2_eligible(dtype, p.M, p.N1, p.N2, p.M)))) {
        dim3 gridw(tn_w2_schedule(q, cdiv(p.N1, 128), cdiv(p.N2, 256), 1));
        hipLaunchKernelGGL(gemm_tn_w2_kernel<true>, gridw, dim3(256), 0, s, q);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (c2d) {
        dim3 grid(cdiv(p.N1, 128) * cdiv(p.N2, 128) * p.splitk);
        if (dtype == 1) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, true, true>), grid, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((gemm_tn_kernel<float, false, true>), grid, dim3(256), 0, s, q);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (p.use_tr && (gemm_tn_uses_w2(dtype, p.M, p.N1, p.N2, p.Tlen) || (p.force_w2 && gemm_tn_w2_eligible(dtype, p.M, p.N1, p.N2, p.Tlen)))) {
        dim3 gridw(tn_w2_schedule(q, cdiv(p.N1, 128), cdiv(p.N2, 256), p.taps));
        hipLaunchKernelGGL(gemm_tn_w2_kernel<false>, gridw, dim3(256), 0, s, q);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    dim3 grid(cdiv(p.N1, 128) * cdiv(p.N2, 128) * p.taps * p.splitk);
    if (dtype == 1) {
        if (p.use_tr) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, true>), grid, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, false>), grid, dim3(256), 0, s, q);
    } else {
        hipLaunchKernelGGL((gemm_tn_kernel<float, false>), grid, dim3(256), 0, s, q);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
