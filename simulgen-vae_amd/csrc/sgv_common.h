// Shared device/host declarations for libsgvae (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define SGV_WAVE 64

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float>  { static constexpr int EPC = 4; };   // elements per 16-byte chunk
template <> struct ElemTraits<bf16_t> { static constexpr int EPC = 8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE (v_cvt_pk_bf16_f32)

// 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for fp32); pointers must be 16-byte aligned.
__device__ __forceinline__ void load8(const float* p, float v[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float v[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}

// 8 consecutive elements as loaded (16 B of bf16 / 32 B of fp32): lets a thread issue several loads before unpacking any
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { bf16x8 v; };
template <> struct Raw8<float> { float4 a, b; };
__device__ __forceinline__ void raw_load(const bf16_t* p, Raw8<bf16_t>& r) { r.v = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void raw_load(const float* p, Raw8<float>& r) {
    r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4);
}
__device__ __forceinline__ void raw_unpack(const Raw8<bf16_t>& r, float v[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r.v[i];
}
__device__ __forceinline__ void raw_unpack(const Raw8<float>& r, float v[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
__device__ __forceinline__ void raw_zero(Raw8<bf16_t>& r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (bf16_t)0.f;
}
__device__ __forceinline__ void raw_zero(Raw8<float>& r) { r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a; }
__device__ __forceinline__ void raw_store(bf16_t* p, const Raw8<bf16_t>& r) { *reinterpret_cast<bf16x8*>(p) = r.v; }
__device__ __forceinline__ void raw_store(float* p, const Raw8<float>& r) {
    *reinterpret_cast<float4*>(p) = r.a; *reinterpret_cast<float4*>(p + 4) = r.b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// erf-GELU (nn.GELU default) and derivative.  Phi(x) = 0.5 (1 + erf(x / sqrt 2)) through Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 absolute, i.e. fp32 rounding level) instead of libm erff: one exp + one fast divide + five FMAs, and
// the exponential exp(-x^2/2) is the one the derivative's density term needs anyway.  The small GroupNorm+GELU layers are
// VALU-bound on this function (libm erff + expf were ~2/3 of their instructions).
__device__ __forceinline__ void gelu_parts(float x, float& Phi, float& dens) {
    const float e = __expf(-0.5f * x * x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678118654752f * fabsf(x));   // v_rcp_f32 (1 ulp); hipcc
    // expands __fdividef to the full IEEE sequence (div_scale / fmas / fixup + Newton steps: ~10 VALU ops per element)
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float h = 0.5f - 0.5f * poly * e;                // 0.5 * erf(|x| / sqrt 2)
    Phi = x >= 0.f ? 0.5f + h : 0.5f - h;
    dens = e * 0.39894228040143268f;
}
__device__ __forceinline__ float gelu_f(float x) { float P, d; gelu_parts(x, P, d); return x * P; }
// tanh through one exp and one v_rcp_f32 (absolute error ~2e-7; libm tanhf made the recon-head passes VALU-bound)
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
__device__ __forceinline__ float gelu_grad_f(float x) { float P, d; gelu_parts(x, P, d); return P + x * d; }

// ---- Philox4x32-10 (counter-based RNG; keyed by (seed, stream), counter = element index / 4) ----
__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// 4 standard normals from one Philox block (Box-Muller)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t stream, uint64_t idx4, float out[4]) {
    uint32_t c[4] = {(uint32_t)idx4, (uint32_t)(idx4 >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float s = 2.3283064365386963e-10f;  // 2^-32
    const float u0 = ((float)c[0] + 0.5f) * s, u1 = ((float)c[1] + 0.5f) * s;
    const float u2 = ((float)c[2] + 0.5f) * s, u3 = ((float)c[3] + 0.5f) * s;
    const float r0 = sqrtf(-2.0f * __logf(u0)), r1 = sqrtf(-2.0f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    out[0] = r0 * c0; out[1] = r0 * s0; out[2] = r1 * c1; out[3] = r1 * s1;
}

// ---------------- GEMM parameter blocks (see gemm.hip) ----------------
struct GemmNT {
    const void* A; long lda;        // activations [M][lda], K columns used
    const void* W; long ldw;        // weights [taps][N][ldw] (K contiguous)
    long w_tap_stride;              // elements between taps
    void* C; long ldc;              // output [M][ldc]
    const void* addend; long ldadd; // optional, same dtype as the GEMM element type
    const float* bias;              // optional [N]
    const float* scale;             // optional device scalar (1/sigma)
    float* partial;                 // split-K slabs [splitk][M][N] fp32
    int M, N, K, taps, pad, Tlen, splitk, out_f32;
    long a_bytes, w_bytes;          // filled by launch_gemm_nt: extents for the buffer descriptors
    // optional: GroupNorm statistics of the output, accumulated by the epilogue of the 128x128 bf16 kernel (splitk 1,
    // bf16 output, Tlen >= 128, gn_Cg >= 128; see gemm_nt_can_fuse_stats): gn_sums[(m / Tlen) * gn_G + n / gn_Cg][2] +=
    // (sum, sum of squares) of the stored (bf16-rounded) values, fp64 atomics into a zeroed buffer
    double* gn_sums; int gn_Cg, gn_G;
    // gemm256.hip (256x256 persistent kernel): deterministic GroupNorm statistics.  gn_part != null: every (work item, wave)
    // writes 8 floats (sum / sum of squares of all, of the rows of the next sample, of the columns of the next group, of
    // both) to gn_part[(item * 8 + wave) * 8 ..]; t256_stats_finalize then produces gn_sums in a fixed order (no atomics).
    float* gn_part;
    long a_rows;                    // rows of the A buffer that may be read (>= M; 0: M) -- lets a launch cover rows [0, M) of a
                                    // longer batch whose taps reach into the rows after M (launch_gemm_nt256 main + tail split)
    int row0;                       // first row this launch computes (128x128 / 128x256 kernels; tiles start at row0)
    // 2-D taps (cv_kw > 0; the latent conditioner's convolutions as implicit GEMMs): A is a channels-last image batch
    // [B][cv_H][cv_W][lda], row m = (b, oh, ow) of a [B][cv_Ho][cv_Wo] output, tap j = (kh, kw) = (j / cv_kw, j % cv_kw) reads
    // A[b][oh*cv_S - cv_P + kh][ow*cv_S - cv_P + kw][k] (zero outside the image); taps = KH * cv_kw; pad / Tlen are not used
    // and a_rows must be B*cv_H*cv_W.  cv_flip: tap j multiplies W[taps-1-j] (the input gradient of a stride-1 convolution
    // is the convolution of dY with the taps reversed).  128x128 bf16/fp32 kernel and the 256x256 kernel only.
    int cv_kw, cv_H, cv_W, cv_S, cv_P, cv_Ho, cv_Wo, cv_flip;
    // strided addend (add_W > 0; bf16 128-row kernel, split-K 1): rows are the pixels of [B][add_H][add_W] images and `addend` is a
    // half-resolution batch [B][ceil(add_H/2)][ceil(add_W/2)][ldadd] that is added at the even pixels only -- the input gradient
    // of a stride-2 1x1 convolution folded into the GEMM that produces the other branch's input gradient
    int add_H, add_W;
    // gemm256.hip item order, filled by launch_gemm_nt256: band > 0 = row tiles in bands of `band` (band-major list);
    // strm = cache policy of the weight / output streams (bit 0: non-temporal weight loads, bit 1: sc1 output stores)
    int band, strm;
    int ts;                         // tile shape of the 256-thread... 512-thread persistent kernel: 0 = 256 x 256, 1 = 128 x 512 (set by
                                    // launch_gemm_nt_planned from the plan; tests set it directly); < 0: the planner must not pick 128 x 512
    int trow0;                      // gemm256.hip: row 0 of this launch is row trow0 of the sample grid (a launch on a row range addressed by shifted
                                    // pointers: the tap windows of a multi-tap product follow the absolute row)
};
// gemm256.hip
bool gemm_nt256_eligible(int dtype, const GemmNT& p);
int gemm_nt256_pick_splitk(int M, int N, int K, int taps);
size_t gemm_nt256_part_floats(int M, int N, int splitk);
int launch_gemm_nt256(const GemmNT& p, hipStream_t s);
// Kernel choice for one NT GEMM / implicit-GEMM convolution (gemm256.hip): kind 0 = 128x128 / 128x256 kernels of gemm.hip,
// 1 = 256x256 persistent kernel over all rows, 3 = the same kernel with 128 x 512 tiles over all rows (M = 3200 = 25 row tiles:
// no tail), 2 = 256x256 kernel over the first m_main rows (a multiple of 256) + the
// gemm.hip kernels over the remaining <= 128 rows (M = 3200 is 12.5 row tiles: thirteen 256-row tiles would need two rounds
// of the 256 workgroups where 12 x N/256 fit one).  fuse_stats: the GroupNorm statistics can come from the GEMM epilogue
// deterministically (kind 1, split-K 1).
struct GemmPlan { int kind, sk_main, sk_tail, m_main, fuse_stats; };
GemmPlan gemm_nt_plan(int dtype, const GemmNT& p, size_t partial_floats, int want_stats);
// A kind-2 plan (256 x 256 kernel on the first rows + a 128-row tail launch) whose tail can run BESIDE the main launch as one round
// of <= 16 items of the 128 x 512 tile shape (the main launch leaves that many CUs free and the tail is over before it):
// gemm_nt_tail_split returns the tail's split-K factor, 0 if the plan does not qualify; launch_gemm_nt_main / _tail issue the halves.
int gemm_nt_tail_split(int dtype, const GemmNT& p, const GemmPlan& pl, size_t tail_partial_floats);
int launch_gemm_nt_main(const GemmNT& p, const GemmPlan& pl, hipStream_t s);
int launch_gemm_nt_tail(const GemmNT& p, const GemmPlan& pl, int sk_tail, float* tail_partial, hipStream_t s);
int launch_gemm_nt_planned(int dtype, const GemmNT& p, const GemmPlan& pl, hipStream_t s);
bool gemm_nt_can_fuse_stats(int dtype, int M, int N, int K, int taps, int Tlen, int Cg);
struct GemmTN {
    const void* A; long lda;        // dY [M][lda], N1 columns used
    const void* B; long ldb;        // X [M][ldb], N2 columns used, row-shifted by tap
    float* out; long ldo;           // dW [taps][N1][ldo] fp32 (splitk > 1: slab z at out + z*out_slab_stride)
    long out_tap_stride;
    long out_slab_stride;
    int M, N1, N2, taps, pad, Tlen, splitk, use_tr;
    int force_w2;                   // tests: take gemm_tn_w2_kernel whenever the shape is eligible (ignores SGV_TN_W2); 2: and its persistent walk;
                                    // 3: the 256 x 256 kernel (gemm256tn.hip) whenever eligible; -1: never the 256 x 256 kernel
    long a_bytes, b_bytes;          // filled by launch_gemm_tn
    // virtual im2col operand (cv_kw > 0; weight gradient of a 2-D convolution without the im2col matrix): B is a channels-
    // last image batch [nb][cv_H][cv_W][ldb] with cv_C channels, reduction row m = output pixel (b, oh, ow) of a
    // [nb][cv_Ho][cv_Wo] grid, column n2 = (kh*cv_kw + kw)*cv_C + c reads B[b][oh*cv_S - cv_P + kh][ow*cv_S - cv_P + kw][c]
    // (zero outside the image); N2 = KH*cv_kw*cv_C, taps = 1, pad / Tlen are not used.
    int cv_kw, cv_H, cv_W, cv_S, cv_P, cv_Ho, cv_Wo, cv_C;
    // gemm_tn_w2_kernel schedule, filled by launch_gemm_tn: order 1 = persistent walk over an XCD-chunked item list cut into
    // patches of pt1 x pt2 tiles x all taps (see the kernel); 0 = one item per block (round 1's raster)
    int order, pt1, pt2;
    int* sched;                     // 256 x 256 kernel: 513 ints of device memory for the work-stealing form (gemm256tn.hip, STEAL; zeroed by the
                                    // launcher), or null = static item lists only
    int out_bf16;                   // 256 x 256 kernel, splitk 1 only: `out` is a bf16 array of the same element layout (the gradient goes to
                                    // the optimizer in bf16: 2 B written and 2 B read per parameter instead of 4 + 4)
};

// gemm256tn.hip: 256 x 256 persistent weight-gradient kernel (bf16; same GemmTN contract, fp32 output / split-K slabs)
bool gemm_tn256_eligible(int dtype, const GemmTN& p);
bool gemm_tn_uses_t256(int dtype, const GemmTN& p);       // the launcher's choice: enough items to fill the chip, >= 150 GFLOP
int launch_gemm_tn256(const GemmTN& p, hipStream_t s);
int launch_gemm_nt(int dtype, const GemmNT& p, hipStream_t s);
int launch_gemm_tn(int dtype, const GemmTN& p, hipStream_t s);
int gemm_nt_pick_splitk(int M, int N, int K, int taps, int dtype);
void gemm_nt_main_done_event(hipEvent_t ev);
bool gemm_nt_uses_wide(int dtype, int N, int K, int taps);   // 128x256 software-pipelined kernel vs the 128x128 one
int gemm_tn_pick_splitk(int M, int N1, int N2, int taps, int dtype, int Tlen = 0);
bool gemm_tn_uses_w2(int dtype, int M, int N1, int N2, int Tlen);   // 128x256 LDS-DMA kernel (2 blocks/CU) vs the 128x128 one
