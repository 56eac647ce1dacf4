// Fused small convolution + GroupNorm + GELU (+ residual) stages for gfx950, bf16: the forward kernel described here and its
// backward mirror (conv_gn_bwd_kernel below: input gradient of the upper convolution + GroupNorm / GELU backward of the stage
// below it, with the residual addend, the gelu' pre-multiply and the stored input gradient the block graph needs).
//
//   y[m][n]   = bf16( scale * sum_{tap,k} x[m + tap - pad][k] * W[tap][n][k] + bias[n] )          (stored: backward reads it)
//   out[m][n] = [res[m][n] + rscale *] gelu( gamma[n] * (y - mean_bg) * rstd_bg + beta[n] )
//   sums[b][g] = (sum y, sum y^2) over the T x Cg slab of sample b, group g                        (stored: backward reads it)
//
// Reference call sites: every Conv1d -> GroupNorm(8, C) -> GELU of modules/common.py:78-162 (encoder / decoder blocks and
// residual blocks) and modules/decoder.py:59-117 (condition_z / condition_xz) whose channel count is at most 1280.
//
// Why: for these layers the step spent three launches -- an implicit-GEMM kernel with split-K (few 128-row tiles exist), its
// combine pass and the fused GroupNorm kernel -- of 15 + 6 + 9 us, each latency-bound.  One workgroup per (group, sample) owns
// the whole T x Cg output slab of its normalisation group (T <= 208 rows, Cg <= 160 columns), so the contraction needs no
// split, the statistics are complete inside the workgroup, and y never has to be re-read: one launch, deterministic (fixed
// summation order), 128 workgroups of 8 waves.
//
// Structure: K in chunks of 32; a chunk of the sample's rows (with `pad` halo rows of zeros above and below: the workgroup is
// exactly one sample, so the tap window test is a zero row) and of the group's weight rows for every tap goes HBM/L2 -> LDS by
// LDS-DMA into a ring of 2-4 stages (counted vmcnt waits, one barrier per chunk, XOR-swizzled 64-byte rows: conflict-free
// ds_read_b128 fragments); a tap is a row offset into the SAME LDS chunk, so the activations are fetched once for all taps.
// Eight waves, two per SIMD: wave w owns row tiles w and w + 8 (16 rows each) times all Cg/16 column tiles of
// v_mfma_f32_16x16x32_bf16; the fragments of tap t + 1 are read while tap t is multiplied (two register sets, counted lgkmcnt);
// the MFMA takes the weight fragment as its first operand (a lane ends with four consecutive channels of one row).
#include "sgv_common.h"
#include "sgv_ew.h"

// sum of one double per thread over the 512-thread block in a fixed order (wave butterfly, then the eight waves in order)
__device__ __forceinline__ double cg_block_sum(double v, double* sm16) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm16[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sm16[0] + sm16[1]) + (sm16[2] + sm16[3])) + ((sm16[4] + sm16[5]) + (sm16[6] + sm16[7]));
}
static constexpr int CG_MAX_T = 208;              // 13 row tiles
constexpr int CG_AR = 224;                 // LDS rows of the activation region: 14 blocks of 16 (208 + 4 halo rows, padded)
typedef __attribute__((address_space(3))) void cg_lds_t;

// One K chunk = 32 channels = 64 bytes per LDS row.  Rows are written by LDS-DMA in blocks of 16 (one wave instruction = 64
// lanes x 16 bytes = 16 rows x 4 slots); slot p of row r holds source chunk p ^ ((r >> 2) & 3), so the 16 rows a
// ds_read_b128 fragment touches (same k-group, consecutive rows) fall into 16 different bank groups.
template <int TAPS, int CT> struct CgGeom {
    static constexpr int WR = TAPS * CT * 16;                         // weight rows per stage: [tap][column]
    static constexpr int SB = (CG_AR + WR) * 64;                      // stage bytes
    static constexpr int NS = (147456 / SB) >= 4 ? 4 : ((147456 / SB) >= 3 ? 3 : 2);      // ring depth
    static constexpr int LDS = NS * SB;
};
struct CgOperands { const void* A; long lda; const void* W; long ldw; long w_tap_stride; int T, K, pad, Cg; };
// acc[j][c] (+)= the 16 x 16 tile (row tile wave + 8 j, column tile c) of  sum_{tap,k} A[b*T + m + tap - pad][k] * W[tap][g*Cg + n][k]
// for the workgroup's (group g, sample b); ends with a barrier (the LDS is free again).
template <int TAPS, int CT>
__device__ __forceinline__ void cg_contract(const CgOperands& p, int g, int b, unsigned char* smem, f32x4 (&acc)[2][CT]) {
    constexpr int SB = CgGeom<TAPS, CT>::SB, NS = CgGeom<TAPS, CT>::NS;
    constexpr int NBLK = CG_AR / 16 + TAPS * CT;               // DMA blocks per stage
    constexpr int NI = (NBLK + 7) / 8;                         // DMA instructions per wave and stage (surplus ones repeat the last block)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, q = lane >> 4;
    const int T = p.T, RT = (T + 15) >> 4;
    const int arows = T + TAPS - 1;                                    // LDS row j <-> time j - pad
    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(p.A) + (long)b * T * p.lda;
    const bf16_t* Wb = reinterpret_cast<const bf16_t*>(p.W) + (long)g * p.Cg * p.ldw;
    const int nk = p.K >> 5;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, (int)((((long)T - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(Wb), 0, (int)((((long)TAPS - 1) * p.w_tap_stride + ((long)p.Cg - 1) * p.ldw + p.K) * 2), 0x00020000);

    // ---- DMA plan of this wave: block wave + 8 i; per-lane source offsets are loop constants, the chunk goes to the scalar offset ----
    uint32_t voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int blk = wave + 8 * i;
        if (blk > NBLK - 1) blk = NBLK - 1;
        const int r = (lane >> 2), pslot = lane & 3, c = pslot ^ (lane >> 4);          // (row >> 2) & 3 == lane >> 4 inside a block
        if (blk < CG_AR / 16) {
            const int row = blk * 16 + r, t = row - p.pad;
            voff[i] = (row < arows && t >= 0 && t < T) ? (uint32_t)(((long)t * p.lda + c * 8) * 2) : 0x80000000u;
        } else {
            const int wr = (blk - CG_AR / 16) * 16 + r;
            const int tap = wr / (CT * 16), n = wr - tap * (CT * 16);
            voff[i] = (uint32_t)(((long)tap * p.w_tap_stride + (long)n * p.ldw + c * 8) * 2);
        }
    }
#define CG_ISSUE(KC)                                                                                          \
    {                                                                                                         \
        const int kc_ = (KC);                                                                                 \
        unsigned char* st_ = smem + (kc_ % NS) * SB;                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                                   \
            int blk_ = wave + 8 * i_;                                                                         \
            if (blk_ > NBLK - 1) blk_ = NBLK - 1;                                                             \
            const uint32_t vo_ = voff[i_];     /* a scalar copy: the array element itself as the builtin's argument breaks the host-side instantiation */ \
            if (blk_ < CG_AR / 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (cg_lds_t*)(st_ + blk_ * 1024), 16, vo_, kc_ * 64, 0, 0); \
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (cg_lds_t*)(st_ + blk_ * 1024), 16, vo_, kc_ * 64, 0, 0); \
        }                                                                                                     \
    }

    // ---- fragment addresses (bytes inside a stage): lane part; row tile / column tile / tap go to adds and immediates ----
    const uint32_t smem_b = (uint32_t)(uintptr_t)(cg_lds_t*)smem;
    const uint32_t wlane = CG_AR * 64 + lr * 64 + ((q ^ (lr >> 2)) << 4);
    uint32_t alane[TAPS];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) alane[tap] = (lr + tap) * 64 + ((q ^ (((lr + tap) >> 2) & 3)) << 4);

    // Eight waves, two per SIMD (one's LDS waits sit under the other's MFMAs): wave w owns row tiles w and w + 8 (13 tiles:
    // waves 5-7 own one; their second fragment read is a clamped duplicate so that every wave issues the same number of LDS
    // reads and the counted lgkmcnt waits below hold).  Measured against four waves x four row tiles (less LDS traffic per
    // MFMA, but one wave per SIMD): 62 vs 74 us on 3200 x 1024 x (3 x 1024).
    const int rt0 = wave, rt1 = wave + 8 < RT ? wave + 8 : wave;
    const bool has1 = wave + 8 < RT;
    bf16x8 fwA[CT], fwB[CT], faA[2], faB[2];                // fragment sets A / B: one tap is multiplied while the next is read
#define CG_READ(FW, FA, TAP)                                                                                  \
    {                                                                                                         \
        _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_)                                                     \
            asm volatile("ds_read_b128 %0, %1" : "=v"(FW[c_]) : "v"(sb + wlane + (uint32_t)(((TAP) * CT + c_) * 1024))); \
        asm volatile("ds_read_b128 %0, %1" : "=v"(FA[0]) : "v"(sb + alane[TAP] + (uint32_t)(rt0 * 1024)));   \
        asm volatile("ds_read_b128 %0, %1" : "=v"(FA[1]) : "v"(sb + alane[TAP] + (uint32_t)(rt1 * 1024)));   \
    }
#define CG_MMA(FW, FA)                                                                                        \
    {                                                                                                         \
        _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_) acc[0][c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FW[c_], FA[0], acc[0][c_], 0, 0, 0); \
        if (has1) { _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_) acc[1][c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FW[c_], FA[1], acc[1][c_], 0, 0, 0); } \
    }
    // all reads of the set have returned / all but the CT + 2 reads issued after them
#define CG_PIN(FW) { _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_) asm volatile("" : "+v"(FW[c_])); }      /* uses of the fragments stay behind the wait */
#define CG_WAIT_ALL(FW, FA) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(FA[0]), "+v"(FA[1])); CG_PIN(FW) }
#define CG_WAIT_OLD(FW, FA) { asm volatile("s_waitcnt lgkmcnt(%[cnt])" : "+v"(FA[0]), "+v"(FA[1]) : [cnt] "n"(CT + 2)); CG_PIN(FW) }

#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nk) CG_ISSUE(s0)
    for (int kc = 0; kc < nk; ++kc) {
        // chunk kc has landed in every wave's part of its stage: the (NS - 2) younger chunks may still be in flight
        if (kc + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((NS - 2) * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        // every wave is past the reads of chunk kc - 1: its stage takes chunk kc + NS - 1
        if (kc + NS - 1 < nk) CG_ISSUE(kc + NS - 1)
        const uint32_t sb = smem_b + (uint32_t)((kc % NS) * SB);
        CG_READ(fwA, faA, 0)
        if (TAPS == 1) {
            CG_WAIT_ALL(fwA, faA)
            CG_MMA(fwA, faA)
        } else {
#pragma unroll
            for (int tap = 0; tap < TAPS; tap += 2) {
                if (tap + 1 < TAPS) {
                    CG_READ(fwB, faB, (tap + 1 < TAPS ? tap + 1 : 0))
                    CG_WAIT_OLD(fwA, faA)
                } else {
                    CG_WAIT_ALL(fwA, faA)
                }
                CG_MMA(fwA, faA)
                if (tap + 1 < TAPS) {
                    if (tap + 2 < TAPS) {
                        CG_READ(fwA, faA, (tap + 2 < TAPS ? tap + 2 : 0))
                        CG_WAIT_OLD(fwB, faB)
                    } else {
                        CG_WAIT_ALL(fwB, faB)
                    }
                    CG_MMA(fwB, faB)
                }
            }
        }
    }
#undef CG_READ
#undef CG_MMA
#undef CG_WAIT_ALL
#undef CG_WAIT_OLD
#undef CG_PIN
#undef CG_ISSUE
    __syncthreads();
}

template <int TAPS, int CT>
__global__ __launch_bounds__(512) void conv_gn_fwd_kernel(const ConvGN p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[CgGeom<TAPS, CT>::LDS];
    __shared__ double smd[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, q = lane >> 4;
    const int g = blockIdx.x, b = blockIdx.y;
    const int T = p.T, RT = (T + 15) >> 4;
    f32x4 acc[2][CT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[j][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        CgOperands o = {p.A, p.lda, p.W, p.ldw, p.w_tap_stride, p.T, p.K, p.pad, p.Cg};
        cg_contract<TAPS, CT>(o, g, b, smem, acc);
    }

    // ---- epilogue: y (bf16, stored), statistics of the stored values, normalise + GELU (+ residual) ----
    const float sc = p.scale ? *p.scale : 1.0f;
    const int n_lane = g * p.Cg + q * 4;                                // + c * 16: four consecutive channels
    bf16_t* yb = reinterpret_cast<bf16_t*>(p.y) + (long)b * T * p.ldy;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n_lane + c * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = (wave + 8 * j) * 16 + lr;
            if (wave + 8 * j < RT && m < T) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 pk;
                pk[0] = (bf16_t)(acc[j][c][0] * sc + bv.x); pk[1] = (bf16_t)(acc[j][c][1] * sc + bv.y);
                pk[2] = (bf16_t)(acc[j][c][2] * sc + bv.z); pk[3] = (bf16_t)(acc[j][c][3] * sc + bv.w);
                *reinterpret_cast<bf16x4*>(yb + (long)m * p.ldy + n_lane + c * 16) = pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (float)pk[r];
                    acc[j][c][r] = v;                                    // keep the stored value for the normalisation
                    s1 += v; s2 += v * v;
                }
            }
        }
    }
    const double S = cg_block_sum((double)s1, smd);
    const double SS = cg_block_sum((double)s2, smd);
    if (tid == 0) {
        p.sums[((long)b * p.G + g) * 2 + 0] = S;
        p.sums[((long)b * p.G + g) * 2 + 1] = SS;
    }
    const double cnt = (double)p.Cg * (double)T;
    const double md = S / cnt;
    double var = SS / cnt - md * md;
    if (var < 0.0) var = 0.0;
    const float mean = (float)md, rstd = (float)(1.0 / sqrt(var + 1e-5));
    bf16_t* ob = reinterpret_cast<bf16_t*>(p.out) + (long)b * T * p.ldout;
    const bf16_t* rb = p.res ? reinterpret_cast<const bf16_t*>(p.res) + (long)b * T * p.ldres : nullptr;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const float4 gm = *reinterpret_cast<const float4*>(p.gamma + n_lane + c * 16);
        const float4 bt = *reinterpret_cast<const float4*>(p.beta + n_lane + c * 16);
        const float ka[4] = {rstd * gm.x, rstd * gm.y, rstd * gm.z, rstd * gm.w};
        const float kb[4] = {bt.x - mean * ka[0], bt.y - mean * ka[1], bt.z - mean * ka[2], bt.w - mean * ka[3]};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = (wave + 8 * j) * 16 + lr;
            if (wave + 8 * j < RT && m < T) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 rr = {0, 0, 0, 0};
                if (rb) rr = *reinterpret_cast<const bf16x4*>(rb + (long)m * p.ldres + n_lane + c * 16);
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float f = gelu_f(acc[j][c][r] * ka[r] + kb[r]);
                    o[r] = (bf16_t)(rb ? (float)rr[r] + p.rscale * f : f);
                }
                *reinterpret_cast<bf16x4*>(ob + (long)m * p.ldout + n_lane + c * 16) = o;
            }
        }
    }
}

// Backward mirror (see ConvGNBwd): dA = scale * sum_{tap,k} dY_up[m + tap - pad][k] * WcT[tap][n][k] (rounded to bf16, as the
// separate input-gradient GEMM stores it), then the GroupNorm + GELU backward of the (group, sample) slab, all sums in a fixed
// order: dz = dA * rscale * gelu'(z);  s1 = sum gamma dz, s2 = sum gamma dz xhat;  dY = gscale * rstd * (gamma dz - s1/n - xhat s2/n).
template <int TAPS, int CT>
__global__ __launch_bounds__(512) void conv_gn_bwd_kernel(const ConvGNBwd p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[CgGeom<TAPS, CT>::LDS];
    __shared__ float smc[3][8][CT * 16];
    __shared__ float smw[3][8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, q = lane >> 4;
    const int g = blockIdx.x, b = blockIdx.y;
    const int T = p.T, RT = (T + 15) >> 4;
    f32x4 acc[2][CT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[j][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        CgOperands o = {p.A, p.lda, p.W, p.ldw, p.w_tap_stride, p.T, p.K, p.pad, p.Cg};
        cg_contract<TAPS, CT>(o, g, b, smem, acc);
    }
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const float sc = p.scale ? *p.scale : 1.0f;
    const double cnt = (double)p.Cg * (double)T;
    const double S = p.sums[((long)b * p.G + g) * 2 + 0], SS = p.sums[((long)b * p.G + g) * 2 + 1];
    const double md = S / cnt;
    double var = SS / cnt - md * md;
    if (var < 0.0) var = 0.0;
    const float mean = (float)md, rstd = (float)(1.0 / sqrt(var + 1e-5));
    const int n_lane = g * p.Cg + q * 4;                                // + c * 16
    const bf16_t* yb = reinterpret_cast<const bf16_t*>(p.y) + (long)b * T * p.ldy;
    f32x4 yv[2][CT];                                                    // y of this lane's outputs (kept for the second pass)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const float4 gm = *reinterpret_cast<const float4*>(p.gamma + n_lane + c * 16);
        const float4 bt = *reinterpret_cast<const float4*>(p.beta + n_lane + c * 16);
        const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
        float cA[4] = {0.f, 0.f, 0.f, 0.f}, cB[4] = {0.f, 0.f, 0.f, 0.f}, cX[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = (wave + 8 * j) * 16 + lr;
            yv[j][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (wave + 8 * j < RT && m < T) {
                const bf16x4 yy = *reinterpret_cast<const bf16x4*>(yb + (long)m * p.ldy + n_lane + c * 16);
                bf16x4 ad = {0, 0, 0, 0};
                if (p.addend) ad = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.addend) + ((long)b * T + m) * p.ldadd + n_lane + c * 16);
                bf16x4 dav = {0, 0, 0, 0};
                bf16x4 pm = {0, 0, 0, 0};
                if (p.premul) pm = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.premul) + ((long)b * T + m) * p.ldpre + n_lane + c * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float yf = (float)yy[r];
                    float da = (float)(bf16_t)(acc[j][c][r] * sc + (float)ad[r]);
                    if (p.premul) da = (float)(bf16_t)(da * gelu_grad_f((float)pm[r]));
                    dav[r] = (bf16_t)da;
                    const float xh = (yf - mean) * rstd;
                    const float qv = da * p.rscale * gelu_grad_f(xh * gmv[r] + btv[r]);
                    yv[j][c][r] = yf;
                    acc[j][c][r] = qv;
                    cA[r] += qv; cB[r] += qv * xh; cX[r] += xh;
                }
                if (p.da) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.da) + ((long)b * T + m) * p.ldda + n_lane + c * 16) = dav;
            } else {
                acc[j][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        // column sums over the 16 rows of the wave's tiles (lanes that share q), then to LDS for the sum over the waves
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { cA[r] += __shfl_xor(cA[r], o, 64); cB[r] += __shfl_xor(cB[r], o, 64); cX[r] += __shfl_xor(cX[r], o, 64); }
        }
        if (lr == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                smc[0][wave][c * 16 + q * 4 + r] = cA[r];
                smc[1][wave][c * 16 + q * 4 + r] = cB[r];
                smc[2][wave][c * 16 + q * 4 + r] = cX[r];
            }
        }
    }
    __syncthreads();
    // column owners: thread n < Cg adds the eight waves in order; s1 / s2 over the group's columns
    float colA = 0.f, colB = 0.f, colX = 0.f, s1 = 0.f, s2 = 0.f, gown = 0.f;
    const bool owner = tid < p.Cg;
    if (owner) {
#pragma unroll
        for (int w = 0; w < 8; ++w) { colA += smc[0][w][tid]; colB += smc[1][w][tid]; colX += smc[2][w][tid]; }
        gown = p.gamma[g * p.Cg + tid];
        s1 = gown * colA; s2 = gown * colB;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); }
    if (lane == 0) { smw[0][wave] = s1; smw[1][wave] = s2; }
    __syncthreads();
    s1 = (smw[0][0] + smw[0][1]) + smw[0][2];      // Cg <= 160: the owners sit in waves 0-2 (the others wrote zeros)
    s2 = (smw[1][0] + smw[1][1]) + smw[1][2];
    if (tid == 0) {
        p.sums2[((long)b * p.G + g) * 2 + 0] = (double)s1;
        p.sums2[((long)b * p.G + g) * 2 + 1] = (double)s2;
    }
    const float m1 = (float)((double)s1 / cnt), m2 = (float)((double)s2 / cnt);
    if (owner) {
        const long C = (long)p.G * p.Cg;
        float* pt = p.ptot + (long)b * 3 * C + g * p.Cg + tid;
        pt[0] = colA;
        pt[C] = colB;
        pt[2 * C] = p.gscale * rstd * (gown * colA - (float)T * m1 - m2 * colX);
    }
    // dY and the <G, W_eff> partial
    bf16_t* dyb = reinterpret_cast<bf16_t*>(p.dy) + (long)b * T * p.lddy;
    float dotacc = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const float4 gm = *reinterpret_cast<const float4*>(p.gamma + n_lane + c * 16);
        const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
        float cb[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.cbias) { const float4 t4 = *reinterpret_cast<const float4*>(p.cbias + n_lane + c * 16); cb[0] = t4.x; cb[1] = t4.y; cb[2] = t4.z; cb[3] = t4.w; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = (wave + 8 * j) * 16 + lr;
            if (wave + 8 * j < RT && m < T) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float yf = yv[j][c][r];
                    const float xh = (yf - mean) * rstd;
                    const float d = rstd * (gmv[r] * acc[j][c][r] - m1 - xh * m2) * p.gscale;
                    dotacc += d * (yf - cb[r]);
                    o[r] = (bf16_t)d;
                }
                *reinterpret_cast<bf16x4*>(dyb + (long)m * p.lddy + n_lane + c * 16) = o;
            }
        }
    }
    if (p.cdot_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dotacc += __shfl_down(dotacc, o, 64);
        if (lane == 0) smw[2][wave] = dotacc;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += smw[2][w];
            p.cdot_part[(long)b * p.G + g] = tot;
        }
    }
}

template <int TAPS>
static int launch_taps_bwd(const ConvGNBwd& p, hipStream_t s) {
    const dim3 grid(p.G, p.B), block(512);
    switch (p.Cg >> 4) {
        case 1: hipLaunchKernelGGL((conv_gn_bwd_kernel<TAPS, 1>), grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_gn_bwd_kernel<TAPS, 2>), grid, block, 0, s, p); break;
        case 4: hipLaunchKernelGGL((conv_gn_bwd_kernel<TAPS, 4>), grid, block, 0, s, p); break;
        case 5: hipLaunchKernelGGL((conv_gn_bwd_kernel<TAPS, 5>), grid, block, 0, s, p); break;
        case 8: hipLaunchKernelGGL((conv_gn_bwd_kernel<TAPS, 8>), grid, block, 0, s, p); break;
        case 10: hipLaunchKernelGGL((conv_gn_bwd_kernel<TAPS, 10>), grid, block, 0, s, p); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int TAPS>
static int launch_taps(const ConvGN& p, hipStream_t s) {
    const dim3 grid(p.G, p.B), block(512);
    switch (p.Cg >> 4) {
        case 1: hipLaunchKernelGGL((conv_gn_fwd_kernel<TAPS, 1>), grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_gn_fwd_kernel<TAPS, 2>), grid, block, 0, s, p); break;
        case 4: hipLaunchKernelGGL((conv_gn_fwd_kernel<TAPS, 4>), grid, block, 0, s, p); break;
        case 5: hipLaunchKernelGGL((conv_gn_fwd_kernel<TAPS, 5>), grid, block, 0, s, p); break;
        case 8: hipLaunchKernelGGL((conv_gn_fwd_kernel<TAPS, 8>), grid, block, 0, s, p); break;
        case 10: hipLaunchKernelGGL((conv_gn_fwd_kernel<TAPS, 10>), grid, block, 0, s, p); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// shapes the fused kernel takes (everything else stays on the GEMM + GroupNorm kernels)
bool conv_gn_fused_eligible(int dtype, const ConvGN& p) {
    if (dtype != 1) return false;
    if (p.taps != 1 && p.taps != 3 && p.taps != 5) return false;
    if (p.pad != (p.taps - 1) / 2) return false;
    if (p.T < 1 || p.T > CG_MAX_T || p.B < 1 || p.G < 1) return false;
    if (p.Cg != 16 && p.Cg != 32 && p.Cg != 64 && p.Cg != 80 && p.Cg != 128 && p.Cg != 160) return false;
    if (p.N != p.G * p.Cg) return false;
    if (p.K < 32 || p.K % 32) return false;
    if (p.lda % 8 || p.ldw % 8 || p.w_tap_stride % 8 || p.ldy % 4 || p.ldout % 4 || (p.res && p.ldres % 4)) return false;
    if (((uintptr_t)p.A | (uintptr_t)p.W) & 15) return false;
    if (((uintptr_t)p.y | (uintptr_t)p.out | (uintptr_t)p.res) & 7) return false;
    if (((uintptr_t)p.bias | (uintptr_t)p.gamma | (uintptr_t)p.beta) & 15) return false;
    return true;
}

int launch_conv_gn_fwd(const ConvGN& p, hipStream_t s) {
    if (!conv_gn_fused_eligible(1, p)) return -1;
    if (p.taps == 1) return launch_taps<1>(p, s);
    if (p.taps == 3) return launch_taps<3>(p, s);
    return launch_taps<5>(p, s);
}

bool conv_gn_bwd_eligible(int dtype, const ConvGNBwd& p) {
    if (dtype != 1) return false;
    if (p.taps != 1 && p.taps != 3 && p.taps != 5) return false;
    if (p.pad != (p.taps - 1) / 2) return false;
    if (p.T < 1 || p.T > CG_MAX_T || p.B < 1 || p.G < 1) return false;
    if (p.Cg != 16 && p.Cg != 32 && p.Cg != 64 && p.Cg != 80 && p.Cg != 128 && p.Cg != 160) return false;
    if (p.N != p.G * p.Cg) return false;
    if (p.K < 32 || p.K % 32) return false;
    if (p.lda % 8 || p.ldw % 8 || p.w_tap_stride % 8 || p.ldy % 4 || p.lddy % 4) return false;
    if (((uintptr_t)p.A | (uintptr_t)p.W) & 15) return false;
    if (((uintptr_t)p.y | (uintptr_t)p.dy | (uintptr_t)p.addend | (uintptr_t)p.premul | (uintptr_t)p.da) & 7) return false;
    if ((p.addend && p.ldadd % 4) || (p.premul && p.ldpre % 4) || (p.da && p.ldda % 4)) return false;
    if (((uintptr_t)p.gamma | (uintptr_t)p.beta | (uintptr_t)p.cbias) & 15) return false;
    if (!p.sums || !p.sums2 || !p.ptot) return false;
    return true;
}

int launch_conv_gn_bwd(const ConvGNBwd& p, hipStream_t s) {
    if (!conv_gn_bwd_eligible(1, p)) return -1;
    if (p.taps == 1) return launch_taps_bwd<1>(p, s);
    if (p.taps == 3) return launch_taps_bwd<3>(p, s);
    return launch_taps_bwd<5>(p, s);
}
