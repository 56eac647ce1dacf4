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
template <typename T, int KCH>
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
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + 127) >> 7, tiles_m = (p.M + 127) >> 7;
    const int ntiles = tiles_n * tiles_m;
    // 1-D grid of ntiles*splitk blocks: XCD-chunked, split-K slice slowest, grouped tile raster within.
    const int logical = xcd_remap(blockIdx.x, ntiles * p.splitk);
    const int z = logical / ntiles;
    const int tile = logical - z * ntiles;
    int tm, tn;
    grouped_raster(tile, tiles_m, tiles_n, tm, tn);
    const int m0 = tm << 7, n0 = tn << 7;
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
    const bool wok0 = wn0 < p.N, wok1 = wn1 < p.N;
    const int at0 = am0 % p.Tlen, at1 = am1 % p.Tlen;
    constexpr int ESZ = (int)sizeof(T);
    const uint32_t abase0 = (uint32_t)(((long)am0 * p.lda + kq * EPC) * ESZ), abase1 = (uint32_t)(((long)am1 * p.lda + kq * EPC) * ESZ);
    const uint32_t wbase0 = wok0 ? (uint32_t)(((long)wn0 * p.ldw + kq * EPC) * ESZ) : OOB_OFF;
    const uint32_t wbase1 = wok1 ? (uint32_t)(((long)wn1 * p.ldw + kq * EPC) * ESZ) : OOB_OFF;
    // two register sets (A, B): loads for tile s+2 are issued while tile s is multiplied and tile s+1 is
    // still in flight (prefetch distance 2; the compiler's in-order vmcnt(4) retires only the older set)
    uint4 ra0A, ra1A, rw0A, rw1A, ra0B, ra1B, rw0B, rw1B;

#define SGV_NT_GLOAD(S, X)                                                                                    \
    {                                                                                                         \
        /* channel chunk outer, tap inner: the taps of one chunk re-read the same activation rows (shifted */ \
        /* by +-2) while they are still in L1/L2; tap-outer order re-streamed the whole panel per tap      */ \
        const int kci_ = (S) / p.taps;                                                                        \
        const int j_ = (S) - kci_ * p.taps;                                                                   \
        const int kc_ = kci_ * BK;                                                                            \
        const int dt_ = j_ - p.pad;                                                                           \
        const bool kok_ = (kc_ + kq * EPC) < p.K;                                                             \
        const int aoff_ = (int)(((long)dt_ * p.lda + kc_) * ESZ);                                             \
        const int woff_ = (int)(((long)j_ * p.w_tap_stride + kc_) * ESZ);                                     \
        const bool pa0 = aok0 && kok_ && ((unsigned)(at0 + dt_) < (unsigned)p.Tlen);                          \
        const bool pa1 = aok1 && kok_ && ((unsigned)(at1 + dt_) < (unsigned)p.Tlen);                          \
        ra0##X = bload16(rsA, pa0 ? abase0 + (uint32_t)aoff_ : OOB_OFF);                                      \
        ra1##X = bload16(rsA, pa1 ? abase1 + (uint32_t)aoff_ : OOB_OFF);                                      \
        rw0##X = bload16(rsW, kok_ ? wbase0 + (uint32_t)woff_ : OOB_OFF);                                     \
        rw1##X = bload16(rsW, kok_ ? wbase1 + (uint32_t)woff_ : OOB_OFF);                                     \
    }
#define SGV_NT_ST1(PTR, V)                                                                                    \
    if constexpr (IS_BF16) { *reinterpret_cast<uint4*>(PTR) = (V); }                                          \
    else { uint32_t* d_ = reinterpret_cast<uint32_t*>(PTR); d_[0] = (V).x; d_[1] = (V).y; d_[2] = (V).z; d_[3] = (V).w; }
#define SGV_NT_SSTORE(BUF, X)                                                                                 \
    {                                                                                                         \
        unsigned char* sa_ = smem + (BUF) * 2 * TILEB + r0 * ROWB + kq * 16;                                  \
        unsigned char* sw_ = sa_ + TILEB;                                                                     \
        SGV_NT_ST1(sa_, ra0##X) SGV_NT_ST1(sa_ + RSTEP * ROWB, ra1##X)                                        \
        SGV_NT_ST1(sw_, rw0##X) SGV_NT_ST1(sw_ + RSTEP * ROWB, rw1##X)                                        \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int a_frag_off = (wm * 64 + lr) * ROWB;
    const int w_frag_off = (wn * 64 + lr) * ROWB;
#define SGV_NT_COMPUTE(BUF)                                                                                   \
    {                                                                                                         \
        const unsigned char* sa_ = smem + (BUF) * 2 * TILEB + a_frag_off;                                     \
        const unsigned char* sw_ = smem + (BUF) * 2 * TILEB + TILEB + w_frag_off;                             \
        if constexpr (IS_BF16) {                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                          \
                const bf16x8 a0_ = *reinterpret_cast<const bf16x8*>(sa_ + (ks * 2 + lh) * 16);                \
                const bf16x8 a1_ = *reinterpret_cast<const bf16x8*>(sa_ + 32 * ROWB + (ks * 2 + lh) * 16);    \
                const bf16x8 b0_ = *reinterpret_cast<const bf16x8*>(sw_ + (ks * 2 + lh) * 16);                \
                const bf16x8 b1_ = *reinterpret_cast<const bf16x8*>(sw_ + 32 * ROWB + (ks * 2 + lh) * 16);    \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b0_, acc[0][0], 0, 0, 0);            \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b1_, acc[0][1], 0, 0, 0);            \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b0_, acc[1][0], 0, 0, 0);            \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b1_, acc[1][1], 0, 0, 0);            \
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
    const bool full = (m0 + 128 <= p.M) && (n0 + 128 <= p.N);
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
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int lcol = wn * 64 + b * 32 + lr;
                    const int gcol = n0 + lcol;
                    const float bv = (p.bias && gcol < p.N) ? p.bias[gcol] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lrow = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        *reinterpret_cast<bf16_t*>(smem + lrow * CP + lcol * 2) = (bf16_t)(acc[a][b][r] * sc + bv);
                    }
                }
            }
            __syncthreads();
            bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = tid + i * 256;          // 2048 chunks of 8 bf16
                const int lrow = c >> 4, lc8 = (c & 15) * 8;
                const int grow = m0 + lrow, gcol = n0 + lc8;
                if (grow < p.M && gcol < p.N) {       // N % 8 == 0: a chunk is entirely in or out
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
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn * 64 + b * 32 + lr;
            const bool cok = full || (col < p.N);
            const int colc = cok ? col : 0;
            const int rbase = m0 + wm * 64 + a * 32 + 4 * lh;
            if (p.splitk > 1) {
                float* dst = p.partial + ((long)z * p.M) * p.N + colc;
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

// =========================================================================================
// NT, LDS-DMA version: operand tiles go global -> LDS directly (buffer_load_dwordx4 ... lds), 3-stage ring,
// prefetch distance 2, one raw s_barrier per K step, no ds_write / staging VGPRs.
//   * LDS image per stage: A[128 rows][64 B] | W[128 rows][64 B], unpadded (an LDS-DMA wave-instruction writes
//     M0-base + lane*16, i.e. 16 consecutive rows).  Bank conflicts are avoided by an XOR swizzle applied on the
//     SOURCE side: LDS slot p of row r holds source chunk p ^ ((r>>2)&3); readers use the same involution.
//   * masked taps / K tails / tile edges: the lane's buffer offset is pointed out of range and the hardware
//     writes zeros into LDS (verified on MI355X: tests/micro/lds_dma_probe.hip).
//   * each wave counts only its own DMA ops: 4 per tile, so `s_waitcnt vmcnt(4)` = "my part of tile s landed,
//     tile s+1 may still be in flight"; the barrier after it publishes all four waves' parts.
// =========================================================================================
typedef __attribute__((address_space(3))) void lds_void;
template <typename T>
__global__ __launch_bounds__(256, SGV_GEMM_MIN_WAVES) void gemm_nt_dma_kernel(const GemmNT p) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = 4 * EPC;
    constexpr bool IS_BF16 = sizeof(T) == 2;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int TILEB = 128 * 64;
    constexpr int STAGEB = 2 * TILEB;
    constexpr int NS = 3;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGEB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + 127) >> 7, tiles_m = (p.M + 127) >> 7;
    const int ntiles = tiles_n * tiles_m;
    const int logical = xcd_remap(blockIdx.x, ntiles * p.splitk);
    const int z = logical / ntiles;
    const int tile = logical - z * ntiles;
    int tm, tn;
    grouped_raster(tile, tiles_m, tiles_n, tm, tn);
    const int m0 = tm << 7, n0 = tn << 7;
    const int kchunks = (p.K + BK - 1) / BK;
    const int total = p.taps * kchunks;
    const int s_begin = (int)((long)total * z / p.splitk);
    const int s_end = (int)((long)total * (z + 1) / p.splitk);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (int)p.w_bytes, 0x00020000);
    // DMA roles: wave w fills tile rows [32w, 32w+32) of each operand with two 1-KiB instructions (16 rows each)
    const int dr0 = wave * 32 + (lane >> 2), dr1 = dr0 + 16;
    const int dp = lane & 3;
    const int dc0 = dp ^ ((dr0 >> 2) & 3), dc1 = dp ^ ((dr1 >> 2) & 3);   // source chunk for this LDS slot
    const int am0 = m0 + dr0, am1 = m0 + dr1, wn0 = n0 + dr0, wn1 = n0 + dr1;
    const bool aok0 = am0 < p.M, aok1 = am1 < p.M;
    const int at0 = am0 % p.Tlen, at1 = am1 % p.Tlen;
    const uint32_t abase0 = (uint32_t)(((long)am0 * p.lda + dc0 * EPC) * ESZ), abase1 = (uint32_t)(((long)am1 * p.lda + dc1 * EPC) * ESZ);
    const uint32_t wbase0 = wn0 < p.N ? (uint32_t)(((long)wn0 * p.ldw + dc0 * EPC) * ESZ) : OOB_OFF;
    const uint32_t wbase1 = wn1 < p.N ? (uint32_t)(((long)wn1 * p.ldw + dc1 * EPC) * ESZ) : OOB_OFF;
    unsigned char* const dmaA = smem + wave * 2048;          // wave-uniform LDS bases (stage 0)
    unsigned char* const dmaW = smem + TILEB + wave * 2048;

#define SGV_DMA_ISSUE(STAGE, S)                                                                               \
    {                                                                                                         \
        const int kci_ = (S) / p.taps;                                                                        \
        const int j_ = (S) - kci_ * p.taps;                                                                   \
        const int kc_ = kci_ * BK;                                                                            \
        const int dt_ = j_ - p.pad;                                                                           \
        const int aoff_ = (int)(((long)dt_ * p.lda + kc_) * ESZ);                                             \
        const int woff_ = (int)(((long)j_ * p.w_tap_stride + kc_) * ESZ);                                     \
        const bool k0_ = (kc_ + dc0 * EPC) < p.K, k1_ = (kc_ + dc1 * EPC) < p.K;                              \
        const bool pa0 = aok0 && k0_ && ((unsigned)(at0 + dt_) < (unsigned)p.Tlen);                           \
        const bool pa1 = aok1 && k1_ && ((unsigned)(at1 + dt_) < (unsigned)p.Tlen);                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(dmaA + (STAGE) * STAGEB), 16, pa0 ? abase0 + (uint32_t)aoff_ : OOB_OFF, 0, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(dmaA + (STAGE) * STAGEB + 1024), 16, pa1 ? abase1 + (uint32_t)aoff_ : OOB_OFF, 0, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(dmaW + (STAGE) * STAGEB), 16, k0_ ? wbase0 + (uint32_t)woff_ : OOB_OFF, 0, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void*)(dmaW + (STAGE) * STAGEB + 1024), 16, k1_ ? wbase1 + (uint32_t)woff_ : OOB_OFF, 0, 0, 0); \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int swz = (lr >> 2) & 3;                           // rows wm*64 + i*32 + lr: bits 2..3 come from lr
    const int a_frag_off = (wm * 64 + lr) * 64;
    const int w_frag_off = TILEB + (wn * 64 + lr) * 64;
#define SGV_DMA_COMPUTE(STAGE)                                                                                \
    {                                                                                                         \
        const unsigned char* sa_ = smem + (STAGE) * STAGEB + a_frag_off;                                      \
        const unsigned char* sw_ = smem + (STAGE) * STAGEB + w_frag_off;                                      \
        if constexpr (IS_BF16) {                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                \
                const int po_ = ((ks * 2 + lh) ^ swz) * 16;                                                   \
                const bf16x8 a0_ = *reinterpret_cast<const bf16x8*>(sa_ + po_);                               \
                const bf16x8 a1_ = *reinterpret_cast<const bf16x8*>(sa_ + 32 * 64 + po_);                     \
                const bf16x8 b0_ = *reinterpret_cast<const bf16x8*>(sw_ + po_);                               \
                const bf16x8 b1_ = *reinterpret_cast<const bf16x8*>(sw_ + 32 * 64 + po_);                     \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b0_, acc[0][0], 0, 0, 0);            \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b1_, acc[0][1], 0, 0, 0);            \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b0_, acc[1][0], 0, 0, 0);            \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b1_, acc[1][1], 0, 0, 0);            \
            }                                                                                                 \
        } else {                                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                \
                const int k_ = ks * 2 + lh;                                                                   \
                const int po_ = (((k_ >> 2) ^ swz) * 16) + (k_ & 3) * 4;                                      \
                const float a0_ = *reinterpret_cast<const float*>(sa_ + po_);                                 \
                const float a1_ = *reinterpret_cast<const float*>(sa_ + 32 * 64 + po_);                       \
                const float b0_ = *reinterpret_cast<const float*>(sw_ + po_);                                 \
                const float b1_ = *reinterpret_cast<const float*>(sw_ + 32 * 64 + po_);                       \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b0_, acc[0][0], 0, 0, 0);               \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b1_, acc[0][1], 0, 0, 0);               \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b0_, acc[1][0], 0, 0, 0);               \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b1_, acc[1][1], 0, 0, 0);               \
            }                                                                                                 \
        }                                                                                                     \
    }

    const int nst = s_end - s_begin;
    if (nst > 0) {
        SGV_DMA_ISSUE(0, s_begin);
        if (nst > 1) SGV_DMA_ISSUE(1, s_begin + 1);
        int st = 0;                                   // stage holding tile i
        int i = 0;
        for (; i + 2 < nst; ++i) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // my 4 DMA ops of tile i have landed
            __builtin_amdgcn_s_barrier();                          // ... and so have the other waves'
            const int st2 = st >= 1 ? st - 1 : st + 2;             // (st + 2) % 3: last read in step i-1
            if (st2 == 0) { SGV_DMA_ISSUE(0, s_begin + i + 2); }
            else if (st2 == 1) { SGV_DMA_ISSUE(1, s_begin + i + 2); }
            else { SGV_DMA_ISSUE(2, s_begin + i + 2); }
            asm volatile("" ::: "memory");
            if (st == 0) { SGV_DMA_COMPUTE(0); } else if (st == 1) { SGV_DMA_COMPUTE(1); } else { SGV_DMA_COMPUTE(2); }
            st = st == 2 ? 0 : st + 1;
        }
        for (; i < nst; ++i) {                                     // last two tiles: nothing left to issue
            if (i + 1 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (st == 0) { SGV_DMA_COMPUTE(0); } else if (st == 1) { SGV_DMA_COMPUTE(1); } else { SGV_DMA_COMPUTE(2); }
            st = st == 2 ? 0 : st + 1;
        }
    }
#undef SGV_DMA_ISSUE
#undef SGV_DMA_COMPUTE

    // ---- epilogue (same as gemm_nt_kernel) ----
    const float sc = p.scale ? *p.scale : 1.0f;
    const bool full = (m0 + 128 <= p.M) && (n0 + 128 <= p.N);
    const T* addp = reinterpret_cast<const T*>(p.addend);
    if constexpr (IS_BF16) {
        if (p.splitk == 1 && !p.out_f32) {
            constexpr int CP = 272;
            static_assert(128 * CP <= NS * STAGEB, "C tile must fit in the ring");
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int lcol = wn * 64 + b * 32 + lr;
                    const int gcol = n0 + lcol;
                    const float bv = (p.bias && gcol < p.N) ? p.bias[gcol] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lrow = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        *reinterpret_cast<bf16_t*>(smem + lrow * CP + lcol * 2) = (bf16_t)(acc[a][b][r] * sc + bv);
                    }
                }
            }
            __syncthreads();
            bf16_t* Cg = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = tid + i * 256;
                const int lrow = c >> 4, lc8 = (c & 15) * 8;
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
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn * 64 + b * 32 + lr;
            const bool cok = full || (col < p.N);
            const int colc = cok ? col : 0;
            const int rbase = m0 + wm * 64 + a * 32 + 4 * lh;
            if (p.splitk > 1) {
                float* dst = p.partial + ((long)z * p.M) * p.N + colc;
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

// split-K combine: out = scale * sum_z partial[z] + bias + addend
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_reduce_kernel(const GemmNT p) {
    const long total = (long)p.M * p.N;
    const float sc = p.scale ? *p.scale : 1.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int row = (int)(i / p.N), col = (int)(i - (long)row * p.N);
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
template <typename T, bool USE_TR>
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
    const int dt = tap - p.pad;
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
    const uint32_t bcol = b_cok ? (uint32_t)((j0 + cq * EPC) * ESZ) : OOB_OFF;
    const uint32_t lda_b = (uint32_t)(p.lda * ESZ), ldb_b = (uint32_t)(p.ldb * ESZ);
    uint4 ra0A, ra1A, rb0A, rb1A, ra0B, ra1B, rb0B, rb1B;

    // rows past M and taps that leave the sample window point out of range -> hardware returns zeros
#define SGV_TN_GLOAD(S, X)                                                                                    \
    {                                                                                                         \
        const int m0_ = (S) * KR + r0, m1_ = (S) * KR + r0 + RSTEP;                                           \
        const bool pa0 = (m0_ < p.M), pa1 = (m1_ < p.M);                                                      \
        const int t0_ = m0_ % p.Tlen, t1_ = m1_ % p.Tlen;                                                     \
        const bool pb0 = pa0 && ((unsigned)(t0_ + dt) < (unsigned)p.Tlen);                                    \
        const bool pb1 = pa1 && ((unsigned)(t1_ + dt) < (unsigned)p.Tlen);                                    \
        ra0##X = bload16(rsA, pa0 ? (uint32_t)m0_ * lda_b + acol : OOB_OFF);                                  \
        ra1##X = bload16(rsA, pa1 ? (uint32_t)m1_ * lda_b + acol : OOB_OFF);                                  \
        rb0##X = bload16(rsB, pb0 ? (uint32_t)(m0_ + dt) * ldb_b + bcol : OOB_OFF);                           \
        rb1##X = bload16(rsB, pb1 ? (uint32_t)(m1_ + dt) * ldb_b + bcol : OOB_OFF);                           \
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
// host launchers
// =========================================================================================
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Split-K choice: minimise (occupancy rounds) x (K steps per block) + the slab combine pass.
// 256 CUs x 3 resident 256-thread blocks (40 KB LDS, ~150 VGPR) = 768 slots; a grid of 800 blocks takes
// two rounds, so e.g. 200 tiles prefer 3 slices (600 blocks) over 4.
static int pick_splitk(long tiles, long steps, double slab_bytes_per_slice, int min_steps) {
    const double slots = 768.0;
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

int gemm_nt_pick_splitk(int M, int N, int K, int taps, int dtype) {
    const int bk = dtype == 1 ? 32 : 16;
    const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
    const long total = (long)taps * cdiv(K, bk);
    return pick_splitk(tiles, total, (double)M * N * 4.0, 8);
}

int gemm_tn_pick_splitk(int M, int N1, int N2, int taps, int dtype) {
    const int kr = dtype == 1 ? 32 : 16;
    const long tiles = (long)cdiv(N1, 128) * cdiv(N2, 128) * taps;
    const long total = cdiv(M, kr);
    return pick_splitk(tiles, total, (double)taps * N1 * N2 * 4.0, 4);
}

int launch_gemm_nt(int dtype, const GemmNT& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return 0;
    const int epc = dtype == 1 ? 8 : 4;
    if (p.K % epc || p.lda % epc || p.ldw % epc || p.w_tap_stride % epc) return -1;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return -1;
    if (p.splitk > 1 && !p.partial) return -1;
    const int esz = dtype == 1 ? 2 : 4;
    GemmNT q = p;
    q.a_bytes = ((long)(p.M - 1) * p.lda + p.K) * esz;
    q.w_bytes = ((long)(p.taps - 1) * p.w_tap_stride + (long)(p.N - 1) * p.ldw + p.K) * esz;
    if (q.a_bytes >= 0x7FFFFFF0L || q.w_bytes >= 0x7FFFFFF0L) return -1;   // 32-bit buffer offsets
    dim3 grid(cdiv(p.M, 128) * cdiv(p.N, 128) * p.splitk);
    static const int use_dma = getenv("SGV_GEMM_DMA") ? atoi(getenv("SGV_GEMM_DMA")) : 0;   // LDS-DMA variant: opt-in (same ~0.65 PF plateau as the register-staged one)
    if (use_dma) {
        if (dtype == 1) hipLaunchKernelGGL((gemm_nt_dma_kernel<bf16_t>), grid, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((gemm_nt_dma_kernel<float>), grid, dim3(256), 0, s, q);
    } else {
        if (dtype == 1) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, 4>), grid, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((gemm_nt_kernel<float, 4>), grid, dim3(256), 0, s, q);
    }
    if (p.splitk > 1) {
        long total = (long)p.M * p.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (dtype == 1) hipLaunchKernelGGL((gemm_nt_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_nt_reduce_kernel<float>), dim3(blocks), dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
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
    if (q.a_bytes >= 0x7FFFFFF0L || q.b_bytes >= 0x7FFFFFF0L) return -1;
    dim3 grid(cdiv(p.N1, 128) * cdiv(p.N2, 128) * p.taps * p.splitk);
    if (dtype == 1) {
        if (p.use_tr) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, true>), grid, dim3(256), 0, s, q);
        else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, false>), grid, dim3(256), 0, s, q);
    } else {
        hipLaunchKernelGGL((gemm_tn_kernel<float, false>), grid, dim3(256), 0, s, q);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
