// Operator-level C ABI for the image latent conditioner (reference
// modules/latent_conditioner_model_cnn.py:28-362): 2-D convolutions as implicit GEMMs on the MFMA kernels of gemm.hip /
// gemm256.hip (2-D tap mode; sgv_op_conv2d_nt / sgv_op_conv2d_tn), the one-channel stem as a direct MFMA convolution with
// its GroupNorm statistics and its weight gradient (stem_conv_*), the im2col / col2im lowering they replace (still the path
// of fp32 stems, stride-2 input gradients and the A/B comparator), GroupNorm(+ReLU) through the kernels of ew.hip incl. the
// one-pass block tail and GroupNorm + ReLU + max-pool, max-pool, squeeze-excitation, residual add+ReLU and the small fp32
// layers of the prediction heads (Linear, LayerNorm, BatchNorm1d, dropout with an injected mask, MSE).
// Feature maps are channels-last [B][H][W][C] (= [B*H*W][C] rows, compute dtype bf16 or fp32), so a 1x1 convolution
// is a plain GEMM and GroupNorm sees the same [rows][C] layout as in the VAE; everything [B][features] is fp32.
// Stateless entry points on caller-owned device buffers; the host-side mirror
// (simulgen-vae_amd/modules/latent_conditioner_model_cnn.py) holds the layer graph and its hand-written backward.
#include "../../include/sgvae_ops.h"
#include "sgv_ew.h"
#include <map>
#include <mutex>
#include <algorithm>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

int sgv_set_error(int code, const char* fmt, ...);   // engine.hip: fills sgv_last_error()

static inline int cdivi(long a, long b) { return (int)((a + b - 1) / b); }
#define OPCHK(cond, ...) do { if (!(cond)) return sgv_set_error(-1, __VA_ARGS__); } while (0)
#define OPLAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : sgv_set_error(-2, "kernel launch failed in %s", __func__))

// ------------------------------------------------------------------------------------------------------------
// im2col / col2im: col[(b,oh,ow)][(kh*KW + kw)*C + c] = x[b][oh*s - p + kh][ow*s - p + kw][c] (0 outside), rows padded
// with zeros to Kp (multiple of 8) elements
// ------------------------------------------------------------------------------------------------------------
struct ConvGeom { int B, H, W, C, KH, KW, S, P, Ho, Wo, Kp; };

// thread = (row lane, k lane): kw_ = min(Kp, 256) k lanes, rows_pb = 256 / kw_ row lanes; a block walks 32 * rows_pb
// consecutive output pixels.  A thread decodes its k once and keeps (b, oh, ow) of its row incrementally (stride
// rows_pb), so the loop body has no divisions and every row of `col` is written as one contiguous run.
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* x, T* col, ConvGeom g, int kw_) {
    const int kl = threadIdx.x % kw_, rl = threadIdx.x / kw_, rows_pb = 256 / kw_;
    const int k = blockIdx.y * kw_ + kl;
    if (k >= g.Kp || rl >= rows_pb) return;
    const bool live = k < g.KH * g.KW * g.C;
    const int c = live ? k % g.C : 0, t = live ? k / g.C : 0, kw = t % g.KW, kh = t / g.KW;
    const long M = (long)g.B * g.Ho * g.Wo;
    long m = (long)blockIdx.x * 32 * rows_pb + rl;
    if (m >= M) return;
    int ow = (int)(m % g.Wo), oh = (int)((m / g.Wo) % g.Ho), b = (int)(m / ((long)g.Wo * g.Ho));
    const T zero = from_f32<T>(0.f);
    for (int pass = 0; pass < 32 && m < M; ++pass, m += rows_pb) {
        T v = zero;
        if (live) {
            const int h = oh * g.S - g.P + kh, w = ow * g.S - g.P + kw;
            if ((unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W) v = x[(((long)b * g.H + h) * g.W + w) * g.C + c];
        }
        col[m * g.Kp + k] = v;
        ow += rows_pb;
        while (ow >= g.Wo) { ow -= g.Wo; if (++oh == g.Ho) { oh = 0; ++b; } }
    }
}
// C == 1 (the stem), stride 1: a thread gathers 8 consecutive k of one output pixel and stores them as one 16-byte (bf16) /
// 32-byte (fp32) piece; consecutive threads -> consecutive pieces, so `col` is written as one contiguous stream
template <typename T>
__global__ __launch_bounds__(256) void im2col_c1_kernel(const T* __restrict__ x, T* __restrict__ col, ConvGeom g) {
    const int nch = g.Kp >> 3;
    const long total = (long)g.B * g.Ho * g.Wo * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / nch;
        const int j = (int)(i - m * nch);
        const int ow = (int)(m % g.Wo), oh = (int)((m / g.Wo) % g.Ho), b = (int)(m / ((long)g.Wo * g.Ho));
        const T* img = x + (long)b * g.H * g.W;
        float v[8];
        int kh = (j * 8) / g.KW, kw = j * 8 - kh * g.KW;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int h = oh * g.S - g.P + kh, w = ow * g.S - g.P + kw;
            v[e] = (kh < g.KH && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W) ? to_f32(img[(long)h * g.W + w]) : 0.f;
            if (++kw == g.KW) { kw = 0; ++kh; }
        }
        store8(col + m * g.Kp + j * 8, v);          // exact: the values are already of type T
    }
}
// C % 8 == 0 (every layer but the 1-channel stem): a lane moves 8 channels (16 bytes for bf16) of one tap; same walk as above
// with kw_ = min(Kp / 8, 256) vector lanes
template <typename T>
__global__ __launch_bounds__(256) void im2col_vec_kernel(const T* x, T* col, ConvGeom g, int kw_) {
    const int kl = threadIdx.x % kw_, rl = threadIdx.x / kw_, rows_pb = 256 / kw_;
    const int kv = blockIdx.y * kw_ + kl;
    if (kv * 8 >= g.Kp || rl >= rows_pb) return;
    const int k = kv * 8;
    const bool live = k < g.KH * g.KW * g.C;
    const int c = live ? k % g.C : 0, t = live ? k / g.C : 0, kw = t % g.KW, kh = t / g.KW;
    const long M = (long)g.B * g.Ho * g.Wo;
    long m = (long)blockIdx.x * 32 * rows_pb + rl;
    if (m >= M) return;
    int ow = (int)(m % g.Wo), oh = (int)((m / g.Wo) % g.Ho), b = (int)(m / ((long)g.Wo * g.Ho));
    for (int pass = 0; pass < 32 && m < M; ++pass, m += rows_pb) {
        Raw8<T> v;
        raw_zero(v);
        if (live) {
            const int h = oh * g.S - g.P + kh, w = ow * g.S - g.P + kw;
            if ((unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W) raw_load(x + (((long)b * g.H + h) * g.W + w) * g.C + c, v);
        }
        raw_store(col + m * g.Kp + k, v);
        ow += rows_pb;
        while (ow >= g.Wo) { ow -= g.Wo; if (++oh == g.Ho) { oh = 0; ++b; } }
    }
}
// ---- stem: one-input-channel convolution, direct on the MFMA (no im2col matrix) -----------------------------------------
// y[b][h][w][n] = scale * sum_{kh,kw} x[b][h - P + kh][w - P + kw] * wp[n][kh*KW + kw]   (stride 1, KH, KW <= 8, bf16)
// A workgroup computes an 8 x 128 pixel tile of one image: the (8 + 8) x (128 + 8) input window sits in LDS (zero outside the
// image), K is laid out as kh*8 + kw (64 slots, weights zero in the unused ones) so a lane's 8 consecutive k of an MFMA
// operand are 8 consecutive pixels of one window row.  Each wave owns 2 rows = 8 tiles of 32 pixels x 32 channels
// (mfma_f32_32x32x16_bf16 x 4); the tile goes through LDS so that every lane stores 16 contiguous bytes, and the wave keeps
// per-channel (sum, sum of squares) of the stored values: per-block partials -> stem_stats_finalize_kernel (fixed order).
constexpr int STEM_TH = 8, STEM_TW = 128, STEM_PITCH = 160, STEM_ROWS = STEM_TH + 8, STEM_SP = 40;
constexpr int STEM_DW_BLOCKS = 1024;      // workgroups (= partial blocks of 2048 floats) of the weight-gradient kernel
__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const float* scale,
                                                            bf16_t* __restrict__ y, float* __restrict__ part, int H, int W, int N, int KH, int KW,
                                                            int P, int Kp) {
    __shared__ __attribute__((aligned(16))) bf16_t img[STEM_ROWS * STEM_PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t stage[4][32 * STEM_SP];
    __shared__ float red[4][32][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w0 = blockIdx.x * STEM_TW, h0 = blockIdx.y * STEM_TH, b = blockIdx.z;
    const bf16_t zero = from_f32<bf16_t>(0.f);
    for (int i = tid; i < STEM_ROWS * (STEM_TW + 8); i += 256) {
        const int r = i / (STEM_TW + 8), c = i - r * (STEM_TW + 8);
        const int h = h0 - P + r, w = w0 - P + c;
        bf16_t v = zero;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) v = x[((long)b * H + h) * W + w];
        img[r * STEM_PITCH + c] = v;
    }
    __syncthreads();
    const float sc = scale ? *scale : 1.f;
    const int ln = lane & 31, lh = lane >> 5;
    const long blk = ((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int n0 = 0; n0 < N; n0 += 32) {
        bf16x8 wf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int kh = 2 * s4 + lh;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                wf[s4][e] = (kh < KH && e < KW && n0 + ln < N) ? wp[(long)(n0 + ln) * Kp + kh * KW + e] : zero;
        }
        float ssum = 0.f, ssq = 0.f;
        for (int t = 0; t < 8; ++t) {
            const int rr = wave * 2 + (t >> 2), tc = (t & 3) * 32;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const bf16_t* src = img + (rr + 2 * s4 + lh) * STEM_PITCH + tc + ln;
                bf16x8 af;
#pragma unroll
                for (int e = 0; e < 8; ++e) af[e] = src[e];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[s4], acc, 0, 0, 0);
            }
            const int h = h0 + rr;
            bf16_t* st = stage[wave];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const bf16_t v = from_f32<bf16_t>(acc[r] * sc);
                if (h < H && w0 + tc + pr < W && n0 + ln < N) { const float f = to_f32(v); ssum += f; ssq += f * f; }
                st[pr * STEM_SP + ln] = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = lane + 64 * j, pr = q >> 2, cc = (q & 3) * 8;
                const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(st + pr * STEM_SP + cc);
                const int w = w0 + tc + pr;
                if (h < H && w < W && n0 + cc < N) *reinterpret_cast<bf16x8*>(y + (((long)b * H + h) * W + w) * N + n0 + cc) = v8;
            }
            __builtin_amdgcn_wave_barrier();
        }
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        __syncthreads();
        if (lane < 32) { red[wave][lane][0] = ssum; red[wave][lane][1] = ssq; }
        __syncthreads();
        if (tid < 32 && n0 + tid < N) {
            float a = 0.f, q2 = 0.f;
            for (int wv = 0; wv < 4; ++wv) { a += red[wv][tid][0]; q2 += red[wv][tid][1]; }
            part[(blk * N + n0 + tid) * 2 + 0] = a;
            part[(blk * N + n0 + tid) * 2 + 1] = q2;
        }
    }
}
// sums[b][g] = (sum, sum of squares) over the image's blocks and the group's channels: one wave per (b, g), lane l takes
// blocks l, l + 64, ... in order, then a shuffle tree -- a fixed order
__global__ __launch_bounds__(64) void stem_stats_finalize_kernel(const float* part, double* sums, int nblk, int N, int G) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int b = i / G, g = i - b * G, Cg = N / G;
    double a = 0.0, q = 0.0;
    for (int k = lane; k < nblk; k += 64)
        for (int c = g * Cg; c < (g + 1) * Cg; ++c) {
            a += (double)part[(((long)b * nblk + k) * N + c) * 2 + 0];
            q += (double)part[(((long)b * nblk + k) * N + c) * 2 + 1];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); q += __shfl_down(q, o, 64); }
    if (lane == 0) { sums[(long)i * 2 + 0] = a; sums[(long)i * 2 + 1] = q; }
}

// Weight gradient of the stem without the im2col matrix: dW[n][kh*8 + kw] = sum_pixels dy[pixel][n] * x[pixel + (kh, kw) - P].
// Same tiling and LDS window as the forward; the reduction runs over 16 consecutive pixels of a row per MFMA
// (A = dy^T: lane -> channel n, 8 pixels, gathered from global memory where 32 lanes read one pixel's 64 contiguous bytes;
// B = window: lane -> slot kh*8 + kw, 8 consecutive pixels of window row kh).  A workgroup walks tiles with a grid stride,
// keeps the 32 x 64 sums of each wave in registers, combines its waves through LDS and writes one partial block;
// stem_dw_finalize_kernel adds the blocks in order into the packed [N][Kp] layout (deterministic, no atomics).
__global__ __launch_bounds__(256) void stem_conv_dw_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ part,
                                                           int B, int H, int W, int N, int n0, int P, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) bf16_t img[STEM_ROWS * STEM_PITCH];
    __shared__ float red[4][2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, lh = lane >> 5;
    const bf16_t zero = from_f32<bf16_t>(0.f);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ntiles = B * tiles_y * tiles_x;
    const bool nok = n0 + ln < N;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (tiles_y * tiles_x), q = t - b * tiles_y * tiles_x;
        const int h0 = (q / tiles_x) * STEM_TH, w0 = (q % tiles_x) * STEM_TW;
        __syncthreads();                                   // the previous tile's window has been consumed
        for (int i = tid; i < STEM_ROWS * (STEM_TW + 8); i += 256) {
            const int r = i / (STEM_TW + 8), c = i - r * (STEM_TW + 8);
            const int h = h0 - P + r, w = w0 - P + c;
            bf16_t v = zero;
            if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) v = x[((long)b * H + h) * W + w];
            img[r * STEM_PITCH + c] = v;
        }
        __syncthreads();
        for (int g = 0; g < 16; ++g) {
            const int rr = wave * 2 + (g >> 3), c0 = (g & 7) * 16 + 8 * lh;
            const int h = h0 + rr;
            bf16x8 af, b0, b1;
            const bf16_t* drow = dy + (((long)b * H + h) * W + w0 + c0) * N + n0 + ln;
#pragma unroll
            for (int e = 0; e < 8; ++e) af[e] = (nok && h < H && w0 + c0 + e < W) ? drow[(long)e * N] : zero;
            const bf16_t* s0 = img + (rr + (ln >> 3)) * STEM_PITCH + c0 + (ln & 7);
            const bf16_t* s1 = s0 + 4 * STEM_PITCH;
#pragma unroll
            for (int e = 0; e < 8; ++e) { b0[e] = s0[e]; b1[e] = s1[e]; }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc1, 0, 0, 0);
        }
    }
    // element (n, slot): n = (r & 3) + 8 * (r >> 2) + 4 * lh, slot = ln (acc0) / 32 + ln (acc1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * lh;
        red[wave][n * 64 + ln] = acc0[r];
        red[wave][n * 64 + 32 + ln] = acc1[r];
    }
    __syncthreads();
    for (int i = tid; i < 2048; i += 256)
        part[(long)blockIdx.x * 2048 + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}
// dW[n0 + n][kh*KW + kw] = sum over the blocks' partials [n][kh*8 + kw]; columns past KH*KW (row padding) are zeroed
// (one wave per output: lane l adds blocks l, l + 64, ... in order, then a shuffle tree -- a fixed order)
__global__ __launch_bounds__(64) void stem_dw_finalize_kernel(const float* part, float* dW, int nblk, int N, int n0, int KH, int KW, int Kp) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int n = i / Kp, kk = i - n * Kp;
    if (n0 + n >= N) return;
    float a = 0.f;
    if (kk < KH * KW) {
        const int kh = kk / KW, kw = kk - kh * KW;
        for (int k = lane; k < nblk; k += 64) a += part[(long)k * 2048 + n * 64 + kh * 8 + kw];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    }
    if (lane == 0) dW[(long)(n0 + n) * Kp + kk] = a;
}

// gather form of col2im with 8 channels per lane: cw_ = min(C / 8, 256) vector lanes
template <typename T>
__global__ __launch_bounds__(256) void col2im_vec_kernel(const T* dcol, T* dx, ConvGeom g, int cw_) {
    const int cl = threadIdx.x % cw_, pl = threadIdx.x / cw_, pix_pb = 256 / cw_;
    const int c = (blockIdx.y * cw_ + cl) * 8;
    if (c >= g.C || pl >= pix_pb) return;
    const long NP = (long)g.B * g.H * g.W;
    for (int pass = 0; pass < 4; ++pass) {
        const long pix = ((long)blockIdx.x * 4 + pass) * pix_pb + pl;
        if (pix >= NP) return;
        const int w = (int)(pix % g.W), h = (int)((pix / g.W) % g.H), b = (int)(pix / ((long)g.W * g.H));
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            const int hn = h + g.P - kh;
            if (hn < 0 || hn % g.S) continue;
            const int oh = hn / g.S;
            if (oh >= g.Ho) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int wn = w + g.P - kw;
                if (wn < 0 || wn % g.S) continue;
                const int ow = wn / g.S;
                if (ow >= g.Wo) continue;
                float v[8];
                load8(dcol + (((long)b * g.Ho + oh) * g.Wo + ow) * g.Kp + (kh * g.KW + kw) * g.C + c, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
        }
        store8(dx + pix * g.C + c, acc);
    }
}
// dx[b][h][w][c] = sum over the windows that cover (h,w) of dcol (gather form: no atomics, deterministic);
// thread = (pixel lane, channel lane) with cw_ = min(C, 256) channel lanes
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* dcol, T* dx, ConvGeom g, int cw_) {
    const int cl = threadIdx.x % cw_, pl = threadIdx.x / cw_, pix_pb = 256 / cw_;
    const int c = blockIdx.y * cw_ + cl;
    if (c >= g.C || pl >= pix_pb) return;
    const long NP = (long)g.B * g.H * g.W;
    for (int pass = 0; pass < 4; ++pass) {
        const long pix = ((long)blockIdx.x * 4 + pass) * pix_pb + pl;
        if (pix >= NP) return;
        const int w = (int)(pix % g.W), h = (int)((pix / g.W) % g.H), b = (int)(pix / ((long)g.W * g.H));
        float acc = 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            const int hn = h + g.P - kh;
            if (hn < 0 || hn % g.S) continue;
            const int oh = hn / g.S;
            if (oh >= g.Ho) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int wn = w + g.P - kw;
                if (wn < 0 || wn % g.S) continue;
                const int ow = wn / g.S;
                if (ow >= g.Wo) continue;
                acc += to_f32(dcol[(((long)b * g.Ho + oh) * g.Wo + ow) * g.Kp + (kh * g.KW + kw) * g.C + c]);
            }
        }
        dx[pix * g.C + c] = from_f32<T>(acc);
    }
}

// ------------------------------------------------------------------------------------------------------------
// MaxPool2d(3, 2, 1): the forward stores the window position (0..8) of each output's first arg-max (PyTorch tie rule:
// first maximum in row-major window order); the backward gathers through it
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pool_argmax(const T* x, const ConvGeom& g, int b, int oh, int ow, int c, float& best, int& bh, int& bw) {
    best = -INFINITY; bh = -1; bw = -1;
    for (int kh = 0; kh < 3; ++kh) {
        const int h = oh * 2 - 1 + kh;
        if ((unsigned)h >= (unsigned)g.H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int w = ow * 2 - 1 + kw;
            if ((unsigned)w >= (unsigned)g.W) continue;
            const float v = to_f32(x[(((long)b * g.H + h) * g.W + w) * g.C + c]);
            if (v > best || bh < 0) { best = v; bh = h; bw = w; }
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* x, T* y, unsigned char* idx, ConvGeom g) {
    const long total = (long)g.B * g.Ho * g.Wo * g.C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        const long m = i / g.C;
        const int ow = (int)(m % g.Wo), oh = (int)((m / g.Wo) % g.Ho), b = (int)(m / ((long)g.Wo * g.Ho));
        float best; int bh, bw;
        pool_argmax(x, g, b, oh, ow, c, best, bh, bw);
        y[i] = from_f32<T>(best);
        if (idx) idx[i] = (unsigned char)((bh - (oh * 2 - 1)) * 3 + (bw - (ow * 2 - 1)));      // window position of the arg-max
    }
}
// 8 channels per lane (C % 8 == 0): same results, 16-byte loads / stores and one 8-byte arg-max store per lane
// coef != null: x is a pre-norm convolution output and every element is first taken through relu(x * ka + kb) with the
// per-(image, channel) GroupNorm coefficients coef[0][b][c] = ka, coef[1][b][c] = kb, rounded to T like the stored
// activation it replaces (sgv_op_gn_relu_maxpool_fwd: the stem's GroupNorm + ReLU + MaxPool in one pass)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_vec_kernel(const T* x, T* y, unsigned char* idx, ConvGeom g, const float* coef = nullptr) {
    const int cv = g.C / 8;
    const long total = (long)g.B * g.Ho * g.Wo * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * 8;
        const long m = i / cv;
        const int ow = (int)(m % g.Wo), oh = (int)((m / g.Wo) % g.Ho), b = (int)(m / ((long)g.Wo * g.Ho));
        float best[8]; int pos[8];
        float ka[8], kb[8];
        if (coef) { load8(coef + (long)b * g.C + c0, ka); load8(coef + ((long)g.B + b) * g.C + c0, kb); }
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; pos[e] = -1; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = oh * 2 - 1 + kh;
            if ((unsigned)h >= (unsigned)g.H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = ow * 2 - 1 + kw;
                if ((unsigned)w >= (unsigned)g.W) continue;
                float v[8];
                load8(x + (((long)b * g.H + h) * g.W + w) * g.C + c0, v);
                if (coef) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = to_f32(from_f32<T>(fmaxf(v[e] * ka[e] + kb[e], 0.f)));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[e] > best[e] || pos[e] < 0) { best[e] = v[e]; pos[e] = kh * 3 + kw; }
            }
        }
        store8(y + m * g.C + c0, best);
        if (idx) {
            unsigned long long pk = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk |= (unsigned long long)(unsigned char)pos[e] << (8 * e);
            *reinterpret_cast<unsigned long long*>(idx + m * g.C + c0) = pk;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_vec_kernel(const unsigned char* idx, const T* dy, T* dx, ConvGeom g) {
    const int cv = g.C / 8;
    const long total = (long)g.B * g.H * g.W * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * 8;
        const long pix = i / cv;
        const int w = (int)(pix % g.W), h = (int)((pix / g.W) % g.H), b = (int)(pix / ((long)g.W * g.H));
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int oh = (h + 1) / 2 - ((h + 1) % 2 == 0 ? 1 : 0); oh <= (h + 1) / 2; ++oh) {
            if (oh < 0 || oh >= g.Ho) continue;
            for (int ow = (w + 1) / 2 - ((w + 1) % 2 == 0 ? 1 : 0); ow <= (w + 1) / 2; ++ow) {
                if (ow < 0 || ow >= g.Wo) continue;
                const long o = (((long)b * g.Ho + oh) * g.Wo + ow) * g.C + c0;
                const unsigned long long pk = *reinterpret_cast<const unsigned long long*>(idx + o);
                const int want = (h - (oh * 2 - 1)) * 3 + (w - (ow * 2 - 1));
                float d[8];
                load8(dy + o, d);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if ((int)((pk >> (8 * e)) & 0xFF) == want) acc[e] += d[e];
            }
        }
        store8(dx + pix * g.C + c0, acc);
    }
}
// gather: an input pixel receives dy of every window whose stored arg-max points at it
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* idx, const T* dy, T* dx, ConvGeom g) {
    const long total = (long)g.B * g.H * g.W * g.C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        const long pix = i / g.C;
        const int w = (int)(pix % g.W), h = (int)((pix / g.W) % g.H), b = (int)(pix / ((long)g.W * g.H));
        float acc = 0.f;
        for (int oh = (h + 1) / 2 - ((h + 1) % 2 == 0 ? 1 : 0); oh <= (h + 1) / 2; ++oh) {
            if (oh < 0 || oh >= g.Ho) continue;
            for (int ow = (w + 1) / 2 - ((w + 1) % 2 == 0 ? 1 : 0); ow <= (w + 1) / 2; ++ow) {
                if (ow < 0 || ow >= g.Wo) continue;
                const long o = (((long)b * g.Ho + oh) * g.Wo + ow) * g.C + c;
                if ((int)idx[o] == (h - (oh * 2 - 1)) * 3 + (w - (ow * 2 - 1))) acc += to_f32(dy[o]);
            }
        }
        dx[i] = from_f32<T>(acc);
    }
}

// ------------------------------------------------------------------------------------------------------------
// elementwise / per-channel helpers on [B][P][C] maps (P = H*W)
// ------------------------------------------------------------------------------------------------------------
// out = relu(a + b) ; backward: d = dout * (out > 0)
// Flat element-wise passes on feature maps: 8 elements per lane (16-byte accesses for bf16) when the length and the
// pointers allow it (n8 = n / 8 vectors), scalar otherwise.  MODE 0: relu(a + b); 1: relu backward (a = output, b = dout);
// 2: a + b
template <typename T, int MODE>
__global__ __launch_bounds__(256) void ew2_kernel(const T* a, const T* b, T* out, long n, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float x[8], y[8];
        load8(a + i * 8, x);
        load8(b + i * 8, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = MODE == 0 ? fmaxf(x[e] + y[e], 0.f) : (MODE == 1 ? (x[e] > 0.f ? y[e] : 0.f) : x[e] + y[e]);
        store8(out + i * 8, x);
    }
    for (long i = n8 * 8 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = to_f32(a[i]), y = to_f32(b[i]);
        out[i] = from_f32<T>(MODE == 0 ? fmaxf(x + y, 0.f) : (MODE == 1 ? (x > 0.f ? y : 0.f) : x + y));
    }
}
static inline long vec8_count(long n, const void* a, const void* b, const void* c) {
    return ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 31) == 0) ? n / 8 : 0;      // 32-byte alignment covers fp32 too
}
// ---- cross-block sums without floating-point atomics ---------------------------------------------------------------------
// A kernel whose blocks each hold a partial of the same output writes it (plain store) to part[r][i] in a per-stream
// workspace; fin_rows_kernel then adds the R partials of every output in index order.  Two launches of the same operator on
// the same inputs give bitwise the same result.
namespace {
struct LcWorkspace { void* p = nullptr; size_t bytes = 0; };
std::mutex g_lc_ws_mu;
std::map<hipStream_t, LcWorkspace> g_lc_ws;
void* lc_workspace(hipStream_t s, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_lc_ws_mu);
    LcWorkspace& w = g_lc_ws[s];
    if (w.bytes < bytes) {
        if (w.p) { hipStreamSynchronize(s); hipFree(w.p); w.p = nullptr; w.bytes = 0; }   // the old buffer may still be read by queued kernels
        const size_t nb = std::max(bytes, (size_t)4 << 20);
        if (hipMalloc(&w.p, nb) != hipSuccess) { w.p = nullptr; return nullptr; }
        w.bytes = nb;
    }
    return w.p;
}
}  // namespace
// out[i] (= | +=) sum_{r < R} part[r * n + i]; one block = CL outputs x (256 / CL) row lanes, lane partials added in lane order
template <typename TP, typename TO, bool ACC, int CL>
__global__ __launch_bounds__(256) void fin_rows_kernel(const TP* part, int R, long n, TO* out) {
    constexpr int RLN = 256 / CL;
    __shared__ double sm[RLN][CL + 1];
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const long i = (long)blockIdx.x * CL + cl;
    double a = 0.0;
    if (i < n) for (int r = rl; r < R; r += RLN) a += (double)part[(long)r * n + i];
    sm[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < RLN; ++k) t += sm[k][cl];
        out[i] = (TO)((ACC ? (double)out[i] : 0.0) + t);
    }
}
template <typename TP, typename TO>
static void fin_rows(const TP* part, int R, long n, TO* out, bool acc, hipStream_t s) {
    if (n >= 64) {
        const dim3 g((unsigned)((n + 63) / 64));
        if (acc) hipLaunchKernelGGL((fin_rows_kernel<TP, TO, true, 64>), g, dim3(256), 0, s, part, R, n, out);
        else hipLaunchKernelGGL((fin_rows_kernel<TP, TO, false, 64>), g, dim3(256), 0, s, part, R, n, out);
    } else {
        const dim3 g((unsigned)n);
        if (acc) hipLaunchKernelGGL((fin_rows_kernel<TP, TO, true, 1>), g, dim3(256), 0, s, part, R, n, out);
        else hipLaunchKernelGGL((fin_rows_kernel<TP, TO, false, 1>), g, dim3(256), 0, s, part, R, n, out);
    }
}
// part[chunk][b][c] = (1/P) sum_{p in chunk} x[b][p][c]   (grid (C/64, B, chunks); 64 channels x 4 row lanes)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* x, float* y, int P, int C, int chunk) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, b = blockIdx.y;
    const int p0 = blockIdx.z * chunk, p1 = min(P, p0 + chunk);
    float a = 0.f;
    if (c < C) for (int p = p0 + rl; p < p1; p += 4) a += to_f32(x[((long)b * P + p) * C + c]);
    __shared__ float sm[4][64];
    sm[rl][threadIdx.x & 63] = a;
    __syncthreads();
    if (rl == 0 && c < C) y[((long)blockIdx.z * gridDim.y + b) * C + c] = (sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]) / (float)P;
}
// dx[b][p][c] (+)= dy[b][c] / P
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_vec_kernel(const float* dy, T* dx, int P, int C, long n8, int accumulate) {
    const int cv = C / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * 8;
        const int b = (int)(i / ((long)P * cv));
        float v[8], o[8];
        load8(dy + (long)b * C + c0, v);
        if (accumulate) load8(dx + i * 8, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = accumulate ? o[e] + v[e] / (float)P : v[e] / (float)P;
        store8(dx + i * 8, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* dy, T* dx, int P, int C, long n, int accumulate) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((long)P * C));
        const float v = dy[(long)b * C + c] / (float)P;
        dx[i] = from_f32<T>(accumulate ? to_f32(dx[i]) + v : v);
    }
}
// out[b][p][c] = x[b][p][c] * s[b][c]
template <typename T>
__global__ __launch_bounds__(256) void chan_scale_fwd_vec_kernel(const T* x, const float* s, T* out, int P, int C, long n8) {
    const int cv = C / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * 8;
        const int b = (int)(i / ((long)P * cv));
        float v[8], sc[8];
        load8(x + i * 8, v);
        load8(s + (long)b * C + c0, sc);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= sc[e];
        store8(out + i * 8, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void chan_scale_fwd_kernel(const T* x, const float* s, T* out, int P, int C, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((long)P * C));
        out[i] = from_f32<T>(to_f32(x[i]) * s[(long)b * C + c]);
    }
}
// dx = dout * s ; part[chunk][b][c] = sum_{p in chunk} dout * x   (grid (C/64, B, chunks))
template <typename T>
__global__ __launch_bounds__(256) void chan_scale_bwd_kernel(const T* x, const float* s, const T* dout, T* dx, float* ds, int P, int C, int chunk) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, b = blockIdx.y;
    const int p0 = blockIdx.z * chunk, p1 = min(P, p0 + chunk);
    float a = 0.f;
    if (c < C) {
        const float sv = s[(long)b * C + c];
        for (int p = p0 + rl; p < p1; p += 4) {
            const long i = ((long)b * P + p) * C + c;
            const float d = to_f32(dout[i]);
            a += d * to_f32(x[i]);
            dx[i] = from_f32<T>(d * sv);
        }
    }
    __shared__ float sm[4][64];
    sm[rl][threadIdx.x & 63] = a;
    __syncthreads();
    if (rl == 0 && c < C) ds[((long)blockIdx.z * gridDim.y + b) * C + c] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------------------------
// small fp32 layers on [B][K] (B <= a few hundred): one wave per output element / row
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float small_act(int act, float z) {
    return act == 1 ? fmaxf(z, 0.f) : (act == 2 ? 1.f / (1.f + __expf(-z)) : z);
}
// y[b][o] = act(scale * sum_k x[b][k] W[o][k] + bias[o]); grid (O, B), one wave
__global__ __launch_bounds__(64) void linear_fwd_kernel(const float* x, const float* W, const float* bias, const float* scale, float* y,
                                                       int K, int O, int act) {
    const int o = blockIdx.x, b = blockIdx.y;
    float a = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) a += x[(long)b * K + k] * W[(long)o * K + k];
    a = wave_sum(a);
    if (threadIdx.x == 0) y[(long)b * O + o] = small_act(act, a * (scale ? *scale : 1.f) + (bias ? bias[o] : 0.f));
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* x, float* y, long n, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = small_act(act, x[i]);
}
// dz = dy * act'(y) (from the stored output y), in place capable
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* y, const float* dy, float* dz, long n, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = y[i];
        dz[i] = dy[i] * (act == 1 ? (v > 0.f ? 1.f : 0.f) : (act == 2 ? v * (1.f - v) : 1.f));
    }
}
// dx[b][k] = scale * sum_o dz[b][o] W[o][k]; grid (ceil(K/64), B): 64 columns x 4 row lanes per block
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* dz, const float* W, const float* scale, float* dx, int K, int O,
                                                           int B, int ochunk) {
    // block = 64 columns k x 8 batch rows x one chunk of outputs: every W element is read once per 8 rows, the chunks run
    // in parallel; each writes its partial to part[chunk][b][k] and fin_rows_kernel adds the chunks in order (onto dx or not)
    constexpr int BG = 8;
    const int kl = threadIdx.x & 63, ol = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl, b0 = blockIdx.y * BG;
    const int o_lo = blockIdx.z * ochunk, o_hi = min(O, o_lo + ochunk);
    float a[BG];
#pragma unroll
    for (int j = 0; j < BG; ++j) a[j] = 0.f;
    if (k < K)
        for (int o = o_lo + ol; o < o_hi; o += 4) {
            const float w = W[(long)o * K + k];
#pragma unroll
            for (int j = 0; j < BG; ++j)
                if (b0 + j < B) a[j] += dz[(long)(b0 + j) * O + o] * w;
        }
    __shared__ float sm[4][BG][64];
#pragma unroll
    for (int j = 0; j < BG; ++j) sm[ol][j][kl] = a[j];
    __syncthreads();
    if (ol == 0 && k < K) {
        const float sc = scale ? *scale : 1.f;
#pragma unroll
        for (int j = 0; j < BG; ++j)
            if (b0 + j < B) dx[((long)blockIdx.z * B + b0 + j) * K + k] = (sm[0][j][kl] + sm[1][j][kl] + sm[2][j][kl] + sm[3][j][kl]) * sc;
    }
}
// dW[o][k] = scale * sum_b dz[b][o] x[b][k]; db[o] = sum_b dz[b][o]; grid (ceil(K/256), O)
// grid (ceil(K/256), ceil(O/8)): a thread keeps its column of x (up to 16 batch rows at a time) in registers and walks 8 outputs,
// so x is read once per 8 outputs instead of once per output; per element the batch rows are added in index order
__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float* __restrict__ dz, const float* __restrict__ x, const float* scale,
                                                           float* __restrict__ dW, float* db, int B, int K, int O) {
    constexpr int OG = 8;
    const int k = blockIdx.x * 256 + threadIdx.x, o0 = blockIdx.y * OG;
    const float sc = scale ? *scale : 1.f;
    if (k < K) {
        float a[OG];
#pragma unroll
        for (int j = 0; j < OG; ++j) a[j] = 0.f;
        for (int b0 = 0; b0 < B; b0 += 16) {
            float xr[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) xr[i] = b0 + i < B ? x[(long)(b0 + i) * K + k] : 0.f;
#pragma unroll
            for (int j = 0; j < OG; ++j) {
                if (o0 + j >= O) break;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (b0 + i < B) a[j] += dz[(long)(b0 + i) * O + o0 + j] * xr[i];
            }
        }
#pragma unroll
        for (int j = 0; j < OG; ++j)
            if (o0 + j < O) dW[(long)(o0 + j) * K + k] = a[j] * sc;
    }
    if (db && blockIdx.x == 0 && threadIdx.x < OG && o0 + (int)threadIdx.x < O) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dz[(long)b * O + o0 + threadIdx.x];
        db[o0 + threadIdx.x] = s;
    }
}
// LayerNorm over the K features of each row (eps 1e-5, biased variance); one wave per row; stores mean/rstd
__global__ __launch_bounds__(64) void layernorm_fwd_kernel(const float* x, const float* gamma, const float* beta, float* y, float* stat, int K) {
    const int b = blockIdx.x;
    float s = 0.f, ss = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) { const float v = x[(long)b * K + k]; s += v; ss += v * v; }
    s = __shfl(wave_sum(s), 0, 64); ss = __shfl(wave_sum(ss), 0, 64);      // wave_sum leaves the total in lane 0
    const float mean = s / K;
    const float var = fmaxf(ss / K - mean * mean, 0.f);
    const float rstd = rsqrtf(var + 1e-5f);
    for (int k = threadIdx.x; k < K; k += 64) y[(long)b * K + k] = (x[(long)b * K + k] - mean) * rstd * gamma[k] + beta[k];
    if (threadIdx.x == 0) { stat[2 * b] = mean; stat[2 * b + 1] = rstd; }
}
// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); per-row terms of dgamma / dbeta to pg[b][k] / pb[b][k] (summed over b by fin_rows_kernel)
__global__ __launch_bounds__(64) void layernorm_bwd_kernel(const float* x, const float* gamma, const float* stat, const float* dy, float* dx,
                                                          float* dgamma, float* dbeta, int K) {
    const int b = blockIdx.x;
    const float mean = stat[2 * b], rstd = stat[2 * b + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) {
        const float xh = (x[(long)b * K + k] - mean) * rstd, g = gamma[k] * dy[(long)b * K + k];
        s1 += g; s2 += g * xh;
    }
    s1 = __shfl(wave_sum(s1), 0, 64) / K; s2 = __shfl(wave_sum(s2), 0, 64) / K;
    for (int k = threadIdx.x; k < K; k += 64) {
        const float xh = (x[(long)b * K + k] - mean) * rstd, d = dy[(long)b * K + k];
        dx[(long)b * K + k] = rstd * (gamma[k] * d - s1 - xh * s2);
        dgamma[(long)b * K + k] = d * xh;
        dbeta[(long)b * K + k] = d;
    }
}
// BatchNorm1d over the batch dimension; train: batch statistics (biased variance for normalisation, unbiased for the
// running buffer, momentum 0.1) ; eval: running statistics.  One thread per feature.  stat[2k], stat[2k+1] = mean, rstd used.
__global__ __launch_bounds__(256) void batchnorm_fwd_kernel(const float* x, const float* gamma, const float* beta, float* run_mean,
                                                           float* run_var, float* y, float* stat, int B, int K, int train) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float mean, var;
    if (train) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += x[(long)b * K + k];
        mean = s / B;
        float v = 0.f;
        for (int b = 0; b < B; ++b) { const float d = x[(long)b * K + k] - mean; v += d * d; }
        var = v / B;
        run_mean[k] = 0.9f * run_mean[k] + 0.1f * mean;
        run_var[k] = 0.9f * run_var[k] + 0.1f * (B > 1 ? v / (B - 1) : var);
    } else { mean = run_mean[k]; var = run_var[k]; }
    const float rstd = rsqrtf(var + 1e-5f);
    for (int b = 0; b < B; ++b) y[(long)b * K + k] = (x[(long)b * K + k] - mean) * rstd * gamma[k] + beta[k];
    stat[2 * k] = mean; stat[2 * k + 1] = rstd;
}
__global__ __launch_bounds__(256) void batchnorm_bwd_kernel(const float* x, const float* gamma, const float* stat, const float* dy, float* dx,
                                                           float* dgamma, float* dbeta, int B, int K, int train) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const float mean = stat[2 * k], rstd = stat[2 * k + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) { const float d = dy[(long)b * K + k], xh = (x[(long)b * K + k] - mean) * rstd; s1 += d; s2 += d * xh; }
    dgamma[k] = s2; dbeta[k] = s1;
    for (int b = 0; b < B; ++b) {
        const float d = dy[(long)b * K + k], xh = (x[(long)b * K + k] - mean) * rstd;
        dx[(long)b * K + k] = train ? gamma[k] * rstd * (d - s1 / B - xh * s2 / B) : gamma[k] * rstd * d;
    }
}
// out = a * mask * keep_scale (dropout with an injected 0/1 mask), also used for its backward
__global__ __launch_bounds__(256) void mask_scale_kernel(const float* a, const float* mask, float scale, float* out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a[i] * (mask ? mask[i] : 1.f) * scale;
}
__global__ __launch_bounds__(256) void addf_kernel(const float* a, const float* b, float* out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a[i] + b[i];
}
// loss = mean((pred - target)^2): block partials to loss[blockIdx.x] (summed by fin_rows_kernel), dpred = gscale * 2 (pred - target) / n
__global__ __launch_bounds__(256) void mse_kernel(const float* pred, const float* target, double* loss, float* dpred, float gscale, long n) {
    __shared__ float smw[4];
    float a = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        a += d * d;
        if (dpred) dpred[i] = gscale * 2.f * d / (float)n;
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) smw[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) loss[blockIdx.x] = ((double)smw[0] + (double)smw[1] + (double)smw[2] + (double)smw[3]) / (double)n;
}
// value of nn.MSELoss / L1Loss / HuberLoss(delta) / SmoothL1Loss(beta) with mean reduction (no gradient): the
// reconstruction term of the end-to-end conditioner loop, which the reference cuts off from the graph
template <int KIND>
__global__ __launch_bounds__(256) void loss_value_kernel(const float* a, const float* b, double* loss, float delta, long n) {
    __shared__ double sm[4];
    double acc = 0.0;
    const long n4 = n >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    auto term = [&](float d) -> float {
        const float ad = fabsf(d);
        if (KIND == 0) return d * d;
        if (KIND == 1) return ad;
        if (KIND == 2) return ad <= delta ? 0.5f * d * d : delta * (ad - 0.5f * delta);
        return ad < delta ? 0.5f * d * d / delta : ad - 0.5f * delta;
    };
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = a4[i], y = b4[i];
        acc += (double)(term(x.x - y.x) + term(x.y - y.y) + term(x.z - y.z) + term(x.w - y.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc += (double)term(a[n4 * 4 + threadIdx.x] - b[n4 * 4 + threadIdx.x]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[blockIdx.x] = (sm[0] + sm[1] + sm[2] + sm[3]) / (double)n;
}
// sklearn MinMaxScaler.inverse_transform on [rows][cols]: (x - min_[c]) / scale_[c]
__global__ __launch_bounds__(256) void cols_sub_div_kernel(const float* x, const float* mn, const float* sc, float* y, long rows, int cols) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cols);
        y[i] = (x[i] - mn[c]) / sc[c];
    }
}
// dtype conversion / layout: [B][C][P] fp32 (reference NCHW with P = H*W) <-> [B][P][C] compute dtype is ew_transpose

// ------------------------------------------------------------------------------------------------------------
// parameter-side helpers: spectral norm (torch legacy hook, modules/common.py:15-37), weight packing, clipping, AdamW
// ------------------------------------------------------------------------------------------------------------
// out = x / max(||x||, eps); one block
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* x, float* out, long n, float eps) {
    __shared__ float sm[4];
    float a = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) a += x[i] * x[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    const float inv = 1.f / fmaxf(sqrtf(sm[0] + sm[1] + sm[2] + sm[3]), eps);
    for (long i = threadIdx.x; i < n; i += 256) out[i] = x[i] * inv;
}
// out4 = {sum a*b, 1/(sum a*b), <8-byte fp64 scratch>}: blocks accumulate into the scratch, dot_final converts
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* a, const float* b, double* acc, long n) {
    __shared__ double sm[4];
    double x = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x += (double)a[i] * (double)b[i];
    x = wave_sum_d(x);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) acc[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void dot_final_kernel(float* out4) {
    const float d = (float)*reinterpret_cast<double*>(out4 + 2);
    out4[0] = d; out4[1] = 1.f / d;
}
// part[rowblock][c] = sum_{r in block} W[r][c] * x[r]   (W^T x; grid (cols/256, rows/64))
__global__ __launch_bounds__(256) void matvec_t_kernel(const float* W, const float* x, float* out, int rows, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * 64, r1 = min(rows, r0 + 64);
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += W[(long)r * cols + c] * x[r];
    out[(long)blockIdx.y * cols + c] = a;
}
// g_orig[r][c] = (G[r][c] - (gw[0] * sig[1]) * u[r] * v[c]) * sig[1]    (sig = {sigma, 1/sigma}, gw[0] = <G, W_orig>)
__global__ __launch_bounds__(256) void sn_grad_kernel(const float* G, const float* u, const float* v, const float* gw, const float* sig,
                                                     float* out, int rows, int cols) {
    const long n = (long)rows * cols;
    const float inv = sig[1], coef = gw[0] * inv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
        out[i] = (G[i] - coef * u[r] * v[c]) * inv;
    }
}
// reference conv weight [Cout][Cin][KH][KW] fp32 <-> GEMM layout [Cout][(kh*KW+kw)*Cin + ci] padded to Kp
template <typename T>
__global__ __launch_bounds__(256) void conv_weight_pack_kernel(const float* w, T* out, int Cout, int Cin, int KH, int KW, int Kp) {
    const long n = (long)Cout * Kp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kp), co = (int)(i / Kp);
        float v = 0.f;
        if (k < KH * KW * Cin) { const int ci = k % Cin, t = k / Cin; v = w[(((long)co * Cin + ci) * KH + t / KW) * KW + t % KW]; }
        out[i] = from_f32<T>(v);
    }
}
__global__ __launch_bounds__(256) void conv_weight_unpack_kernel(const float* packed, float* w, int Cout, int Cin, int KH, int KW, int Kp) {
    const long n = (long)Cout * Cin * KH * KW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int kw = (int)(i % KW), kh = (int)((i / KW) % KH), ci = (int)((i / ((long)KW * KH)) % Cin), co = (int)(i / ((long)KW * KH * Cin));
        w[i] = packed[(long)co * Kp + (kh * KW + kw) * Cin + ci];
    }
}
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long n, double* acc) {
    __shared__ double sm[4];
    double a = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a += (double)g[i] * (double)g[i];
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) acc[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];      // block partial; fin_rows_kernel adds them onto the running total
}
// torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (total_norm + 1e-6)); out = {coef, total_norm}
__global__ void clip_coef_kernel(const double* sumsq, float max_norm, float* out) {
    const float tn = (float)sqrt(sumsq[0]);
    out[0] = fminf(1.f, max_norm / (tn + 1e-6f));
    out[1] = tn;
}
// torch.optim.AdamW step on one tensor; gscale (device scalar, may be NULL) multiplies the gradient first
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                                                        float eps, float wd, float bc1, float bc2s, const float* gscale) {
    const float gs = gscale ? gscale[0] : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * gs;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = m[i] * b1 + (1.f - b1) * gi;
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2s + eps));
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// ------------------------------------------------------------------------------------------------------------
// input augmentation of the latent-conditioner loop (modules/latent_conditioner.py:107-159,261-279) on [B][H][W] fp32
// ------------------------------------------------------------------------------------------------------------
// torch.flip(dims=[2]) where flip[b], then torch.roll by (sx[b] along W, sy[b] along H), in that order
__global__ __launch_bounds__(256) void flip_roll_kernel(const float* x, float* out, int B, int H, int W, const int* flip, const int* sx, const int* sy) {
    const long n = (long)B * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int w = (int)(i % W), h = (int)((i / W) % H), b = (int)(i / ((long)W * H));
        int hs = (h - sy[b]) % H; if (hs < 0) hs += H;          // roll: out[h] = in[h - shift]
        int ws = (w - sx[b]) % W; if (ws < 0) ws += W;
        if (flip[b]) ws = W - 1 - ws;
        out[i] = x[((long)b * H + hs) * W + ws];
    }
}
// F.affine_grid(theta, align_corners=False) + F.grid_sample(bilinear, padding_mode='border', align_corners=False)
__global__ __launch_bounds__(256) void affine_sample_kernel(const float* x, float* out, int B, int H, int W, const float* theta) {
    const long n = (long)B * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int w = (int)(i % W), h = (int)((i / W) % H), b = (int)(i / ((long)W * H));
        const float* t = theta + 6 * b;
        if (t[0] == 1.f && t[1] == 0.f && t[2] == 0.f && t[3] == 0.f && t[4] == 1.f && t[5] == 0.f) {   // sample not selected: untouched
            out[i] = x[i];
            continue;
        }
        const float xn = (2.f * w + 1.f) / W - 1.f, yn = (2.f * h + 1.f) / H - 1.f;      // pixel centres in [-1, 1]
        const float gx = t[0] * xn + t[1] * yn + t[2], gy = t[3] * xn + t[4] * yn + t[5];
        float px = ((gx + 1.f) * W - 1.f) * 0.5f, py = ((gy + 1.f) * H - 1.f) * 0.5f;
        px = fminf(fmaxf(px, 0.f), (float)(W - 1)); py = fminf(fmaxf(py, 0.f), (float)(H - 1));   // border padding
        const int x0 = (int)floorf(px), y0 = (int)floorf(py);
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const float fx = px - x0, fy = py - y0;
        const float* img = x + (long)b * H * W;
        out[i] = (1.f - fy) * ((1.f - fx) * img[(long)y0 * W + x0] + fx * img[(long)y0 * W + x1]) +
                 fy * ((1.f - fx) * img[(long)y1 * W + x0] + fx * img[(long)y1 * W + x1]);
    }
}
// out[b] = lam * x[b] + (1 - lam) * x[perm[b]]   (rows of n floats)
__global__ __launch_bounds__(256) void mixup_rows_kernel(const float* x, const int* perm, float lam, float* out, int B, long n) {
    const long tot = (long)B * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
        const int b = (int)(i / n);
        out[i] = lam * x[i] + (1.f - lam) * x[(long)perm[b] * n + (i - (long)b * n)];
    }
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
static ConvGeom mk_geom(int B, int H, int W, int C, int KH, int KW, int S, int P) {
    ConvGeom g; g.B = B; g.H = H; g.W = W; g.C = C; g.KH = KH; g.KW = KW; g.S = S; g.P = P;
    g.Ho = (H + 2 * P - KH) / S + 1; g.Wo = (W + 2 * P - KW) / S + 1; g.Kp = (KH * KW * C + 7) / 8 * 8;
    return g;
}
static dim3 grid1(long n) { long b = (n + 255) / 256; return dim3((unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b))); }
// launch KERN<T>(args...) for T = bf16 (dtype 1) or float (dtype 0); pointer arguments are cast inside ARGS via PT/CPT
#define ON_DTYPE(DT, ...)                                        \
    do {                                                         \
        if ((DT) == 1) { typedef bf16_t T; __VA_ARGS__; }        \
        else { typedef float T; __VA_ARGS__; }                   \
    } while (0)
#define PT(p) reinterpret_cast<T*>(p)
#define CPT(p) reinterpret_cast<const T*>(p)
#define ST(s) reinterpret_cast<hipStream_t>(s)

static GNParams gn_params(const void* y, int B, int P, int C, int G, const float* gamma, const float* beta, double* sums) {
    GNParams p;
    p.y = y; p.ldy = C; p.gamma = gamma; p.beta = beta; p.sums = sums;
    p.B = B; p.T = P; p.C = C; p.G = G; p.Cg = C / G;
    return p;
}

extern "C" {

int sgv_op_conv_out_shape(int H, int W, int C, int KH, int KW, int stride, int pad, int* Ho, int* Wo, int* Kp) {
    const ConvGeom g = mk_geom(1, H, W, C, KH, KW, stride, pad);
    if (Ho) *Ho = g.Ho;
    if (Wo) *Wo = g.Wo;
    if (Kp) *Kp = g.Kp;
    return 0;
}
// one-input-channel stride-1 convolution + GroupNorm statistics of its output: see include/sgvae_ops.h
size_t sgv_op_stem_conv_workspace_floats(int B, int H, int W, int N) {
    const size_t fwd = (size_t)B * cdivi(H, STEM_TH) * cdivi(W, STEM_TW) * (size_t)N * 2, bwd = (size_t)STEM_DW_BLOCKS * 2048;
    return fwd > bwd ? fwd : bwd;
}
int sgv_op_stem_conv_fwd(const void* x, const void* wp, const float* scale, void* y, double* sums, float* part, int B, int H, int W, int N,
                         int KH, int KW, int pad, int G, void* stream) {
    OPCHK(x && wp && y && sums && part && B > 0 && H > 0 && W > 0 && N > 0, "sgv_op_stem_conv_fwd: bad argument");
    OPCHK(KH >= 1 && KH <= 8 && KW >= 1 && KW <= 8 && pad >= 0 && pad <= 7 && 2 * pad == KH - 1 && KH == KW,
          "sgv_op_stem_conv_fwd: square odd windows up to 7x7 with 'same' padding only (got %dx%d pad %d)", KH, KW, pad);
    OPCHK(N % 8 == 0 && G >= 1 && N % G == 0, "sgv_op_stem_conv_fwd: N %% 8 == 0 and N %% G == 0 required (got N=%d G=%d)", N, G);
    OPCHK(cdivi(H, STEM_TH) <= 65535 && B <= 65535, "sgv_op_stem_conv_fwd: image too tall / batch too large for the grid");
    const int Kp = (KH * KW + 7) / 8 * 8;
    const dim3 grid(cdivi(W, STEM_TW), cdivi(H, STEM_TH), B);
    hipLaunchKernelGGL(stem_conv_fwd_kernel, grid, dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(wp),
                       scale, reinterpret_cast<bf16_t*>(y), part, H, W, N, KH, KW, pad, Kp);
    hipLaunchKernelGGL(stem_stats_finalize_kernel, dim3(B * G), dim3(64), 0, ST(stream), part, sums, (int)(grid.x * grid.y), N, G);
    return OPLAUNCH_OK();
}
int sgv_op_im2col(int dtype, const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int stride, int pad, void* stream) {
    OPCHK(x && col && B > 0 && H > 0 && W > 0 && C > 0 && stride > 0 && KH > 0 && KW > 0, "sgv_op_im2col: bad argument");
    const ConvGeom g = mk_geom(B, H, W, C, KH, KW, stride, pad);
    const long Mrows = (long)B * g.Ho * g.Wo;
    if (C % 8 == 0 && g.Kp % 8 == 0 && ((((uintptr_t)x) | ((uintptr_t)col)) & 31) == 0) {
        const int kv = g.Kp / 8, kwv = kv < 256 ? kv : 256, rpb = 256 / kwv;
        ON_DTYPE(dtype, hipLaunchKernelGGL(im2col_vec_kernel<T>, dim3((unsigned)((Mrows + 32L * rpb - 1) / (32L * rpb)), cdivi(kv, kwv)), dim3(256), 0, ST(stream), CPT(x), PT(col), g, kwv));
        return OPLAUNCH_OK();
    }
    if (C == 1 && g.Kp % 8 == 0 && (((uintptr_t)col) & 31) == 0) {
        const long pieces = Mrows * (g.Kp / 8);
        ON_DTYPE(dtype, hipLaunchKernelGGL(im2col_c1_kernel<T>, dim3((unsigned)std::min<long>((pieces + 255) / 256, 1L << 20)), dim3(256), 0, ST(stream), CPT(x), PT(col), g));
        return OPLAUNCH_OK();
    }
    const int kw_ = g.Kp < 256 ? g.Kp : 256, rows_pb = 256 / kw_;
    ON_DTYPE(dtype, hipLaunchKernelGGL(im2col_kernel<T>, dim3((unsigned)((Mrows + 32L * rows_pb - 1) / (32L * rows_pb)), cdivi(g.Kp, kw_)), dim3(256), 0, ST(stream), CPT(x), PT(col), g, kw_));
    return OPLAUNCH_OK();
}
int sgv_op_col2im(int dtype, const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad, void* stream) {
    OPCHK(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && stride > 0, "sgv_op_col2im: bad argument");
    const ConvGeom g = mk_geom(B, H, W, C, KH, KW, stride, pad);
    const long NPix = (long)B * H * W;
    if (C % 8 == 0 && g.Kp % 8 == 0 && ((((uintptr_t)dcol) | ((uintptr_t)dx)) & 31) == 0) {
        const int cv = C / 8, cwv = cv < 256 ? cv : 256, ppb = 256 / cwv;
        ON_DTYPE(dtype, hipLaunchKernelGGL(col2im_vec_kernel<T>, dim3((unsigned)((NPix + 4L * ppb - 1) / (4L * ppb)), cdivi(cv, cwv)), dim3(256), 0, ST(stream), CPT(dcol), PT(dx), g, cwv));
        return OPLAUNCH_OK();
    }
    const int cw_ = C < 256 ? C : 256, pix_pb = 256 / cw_;
    ON_DTYPE(dtype, hipLaunchKernelGGL(col2im_kernel<T>, dim3((unsigned)((NPix + 4L * pix_pb - 1) / (4L * pix_pb)), cdivi(C, cw_)), dim3(256), 0, ST(stream), CPT(dcol), PT(dx), g, cw_));
    return OPLAUNCH_OK();
}
int sgv_op_maxpool_fwd(int dtype, const void* x, void* y, unsigned char* argmax, int B, int H, int W, int C, void* stream) {
    OPCHK(x && y && B > 0 && H > 0 && W > 0 && C > 0, "sgv_op_maxpool_fwd: bad argument");
    const ConvGeom g = mk_geom(B, H, W, C, 3, 3, 2, 1);
    if (C % 8 == 0 && vec8_count(8, x, y, argmax)) {
        ON_DTYPE(dtype, hipLaunchKernelGGL(maxpool_fwd_vec_kernel<T>, grid1((long)B * g.Ho * g.Wo * (C / 8)), dim3(256), 0, ST(stream), CPT(x), PT(y), argmax, g, (const float*)nullptr));
        return OPLAUNCH_OK();
    }
    ON_DTYPE(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, grid1((long)B * g.Ho * g.Wo * C), dim3(256), 0, ST(stream), CPT(x), PT(y), argmax, g));
    return OPLAUNCH_OK();
}
// ka = rstd * gamma, kb = beta - mean * rstd * gamma per (image, channel) from the GroupNorm statistics (eps 1e-5, as ew.hip)
__global__ __launch_bounds__(256) void gn_coef_kernel(const double* sums, const float* gamma, const float* beta, float* coef, int B, int C, int G, double n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C, g = c / (C / G);
    const double s = sums[((long)b * G + g) * 2 + 0], ss = sums[((long)b * G + g) * 2 + 1];
    const double m = s / n;
    double var = ss / n - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + 1e-5));
    const float ga = gamma[c];
    coef[i] = rstd * ga;
    coef[(long)B * C + i] = beta[c] - mean * rstd * ga;
}
// relu(gn(y)) then MaxPool2d(3, 2, 1) in one pass over y: see include/sgvae_ops.h
int sgv_op_gn_relu_maxpool_fwd(int dtype, const void* y, const double* sums, const float* gamma, const float* beta, int G, void* out,
                               unsigned char* argmax, float* coef, int B, int H, int W, int C, void* stream) {
    OPCHK(y && sums && gamma && beta && out && coef && B > 0 && H > 0 && W > 0 && C > 0, "sgv_op_gn_relu_maxpool_fwd: bad argument");
    OPCHK(C % 8 == 0 && G >= 1 && C % G == 0 && vec8_count(8, y, out, argmax), "sgv_op_gn_relu_maxpool_fwd: C %% 8 == 0, C %% G == 0 and 16-byte aligned buffers required");
    const ConvGeom g = mk_geom(B, H, W, C, 3, 3, 2, 1);
    hipLaunchKernelGGL(gn_coef_kernel, dim3(cdivi((long)B * C, 256)), dim3(256), 0, ST(stream), sums, gamma, beta, coef, B, C, G, (double)(C / G) * H * W);
    ON_DTYPE(dtype, hipLaunchKernelGGL(maxpool_fwd_vec_kernel<T>, grid1((long)B * g.Ho * g.Wo * (C / 8)), dim3(256), 0, ST(stream), CPT(y), PT(out), argmax, g, (const float*)coef));
    return OPLAUNCH_OK();
}
int sgv_op_maxpool_bwd(int dtype, const unsigned char* argmax, const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
    OPCHK(argmax && dy && dx && B > 0 && H > 0 && W > 0 && C > 0, "sgv_op_maxpool_bwd: bad argument");
    const ConvGeom g = mk_geom(B, H, W, C, 3, 3, 2, 1);
    if (C % 8 == 0 && vec8_count(8, argmax, dy, dx)) {
        ON_DTYPE(dtype, hipLaunchKernelGGL(maxpool_bwd_vec_kernel<T>, grid1((long)B * H * W * (C / 8)), dim3(256), 0, ST(stream), argmax, CPT(dy), PT(dx), g));
        return OPLAUNCH_OK();
    }
    ON_DTYPE(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, grid1((long)B * H * W * C), dim3(256), 0, ST(stream), argmax, CPT(dy), PT(dx), g));
    return OPLAUNCH_OK();
}
int sgv_op_add_relu_fwd(int dtype, const void* a, const void* b, void* out, long n, void* stream) {
    OPCHK(a && b && out && n > 0, "sgv_op_add_relu_fwd: bad argument");
    const long n8 = vec8_count(n, a, b, out);
    ON_DTYPE(dtype, hipLaunchKernelGGL((ew2_kernel<T, 0>), grid1(n8 ? n8 : n), dim3(256), 0, ST(stream), CPT(a), CPT(b), PT(out), n, n8));
    return OPLAUNCH_OK();
}
int sgv_op_relu_bwd(int dtype, const void* out, const void* dout, void* d, long n, void* stream) {
    OPCHK(out && dout && d && n > 0, "sgv_op_relu_bwd: bad argument");
    const long n8 = vec8_count(n, out, dout, d);
    ON_DTYPE(dtype, hipLaunchKernelGGL((ew2_kernel<T, 1>), grid1(n8 ? n8 : n), dim3(256), 0, ST(stream), CPT(out), CPT(dout), PT(d), n, n8));
    return OPLAUNCH_OK();
}
int sgv_op_add(int dtype, const void* a, const void* b, void* out, long n, void* stream) {
    OPCHK(a && b && out && n > 0, "sgv_op_add: bad argument");
    const long n8 = vec8_count(n, a, b, out);
    ON_DTYPE(dtype, hipLaunchKernelGGL((ew2_kernel<T, 2>), grid1(n8 ? n8 : n), dim3(256), 0, ST(stream), CPT(a), CPT(b), PT(out), n, n8));
    return OPLAUNCH_OK();
}
int sgv_op_avgpool_fwd(int dtype, const void* x, float* y, int B, int P, int C, void* stream) {
    OPCHK(x && y && B > 0 && P > 0 && C > 0, "sgv_op_avgpool_fwd: bad argument");
    const int chunk = P > 512 ? 256 : P, chunks = cdivi(P, chunk);
    float* part = chunks > 1 ? (float*)lc_workspace(ST(stream), sizeof(float) * (size_t)chunks * B * C) : y;     // one chunk: the block's value is the result
    OPCHK(part, "sgv_op_avgpool_fwd: workspace allocation failed");
    ON_DTYPE(dtype, hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(cdivi(C, 64), B, chunks), dim3(256), 0, ST(stream), CPT(x), part, P, C, chunk));
    if (chunks > 1) fin_rows(part, chunks, (long)B * C, y, false, ST(stream));
    return OPLAUNCH_OK();
}
int sgv_op_avgpool_bwd(int dtype, const float* dy, void* dx, int B, int P, int C, int accumulate, void* stream) {
    OPCHK(dy && dx && B > 0 && P > 0 && C > 0, "sgv_op_avgpool_bwd: bad argument");
    const long n = (long)B * P * C;
    if (C % 8 == 0 && vec8_count(n, dy, dx, dx)) {
        ON_DTYPE(dtype, hipLaunchKernelGGL(avgpool_bwd_vec_kernel<T>, grid1(n / 8), dim3(256), 0, ST(stream), dy, PT(dx), P, C, n / 8, accumulate));
        return OPLAUNCH_OK();
    }
    ON_DTYPE(dtype, hipLaunchKernelGGL(avgpool_bwd_kernel<T>, grid1(n), dim3(256), 0, ST(stream), dy, PT(dx), P, C, n, accumulate));
    return OPLAUNCH_OK();
}
int sgv_op_chan_scale_fwd(int dtype, const void* x, const float* s, void* out, int B, int P, int C, void* stream) {
    OPCHK(x && s && out && B > 0 && P > 0 && C > 0, "sgv_op_chan_scale_fwd: bad argument");
    const long n = (long)B * P * C;
    if (C % 8 == 0 && vec8_count(n, x, s, out)) {
        ON_DTYPE(dtype, hipLaunchKernelGGL(chan_scale_fwd_vec_kernel<T>, grid1(n / 8), dim3(256), 0, ST(stream), CPT(x), s, PT(out), P, C, n / 8));
        return OPLAUNCH_OK();
    }
    ON_DTYPE(dtype, hipLaunchKernelGGL(chan_scale_fwd_kernel<T>, grid1(n), dim3(256), 0, ST(stream), CPT(x), s, PT(out), P, C, n));
    return OPLAUNCH_OK();
}
int sgv_op_chan_scale_bwd(int dtype, const void* x, const float* s, const void* dout, void* dx, float* ds, int B, int P, int C, void* stream) {
    OPCHK(x && s && dout && dx && ds && B > 0 && P > 0 && C > 0, "sgv_op_chan_scale_bwd: bad argument");
    const int chunk = P > 512 ? 256 : P, chunks = cdivi(P, chunk);
    float* part = chunks > 1 ? (float*)lc_workspace(ST(stream), sizeof(float) * (size_t)chunks * B * C) : ds;
    OPCHK(part, "sgv_op_chan_scale_bwd: workspace allocation failed");
    ON_DTYPE(dtype, hipLaunchKernelGGL(chan_scale_bwd_kernel<T>, dim3(cdivi(C, 64), B, chunks), dim3(256), 0, ST(stream), CPT(x), s, CPT(dout), PT(dx), part, P, C, chunk));
    if (chunks > 1) fin_rows(part, chunks, (long)B * C, ds, false, ST(stream));
    return OPLAUNCH_OK();
}

// GroupNorm (+ activation: 0 none, 3 relu) on [B][P][C]; sums: B*G*2 doubles (written); part: sgv_op_gn_workspace_floats floats of
// scratch (per-block partial sums, combined in a fixed order: no atomics)
int sgv_op_gn_fwd(int dtype, int act, const void* y, void* out, int B, int P, int C, int G, const float* gamma, const float* beta,
                  double* sums, float* part, void* stream) {
    OPCHK(y && out && gamma && beta && sums && part, "sgv_op_gn_fwd: null argument");
    OPCHK(C % 8 == 0 && G >= 1 && G <= SGV_GN_MAX_GROUPS && C % G == 0, "sgv_op_gn_fwd: C %% 8 == 0, 1 <= G <= %d, C %% G == 0 required", SGV_GN_MAX_GROUPS);
    GNParams p = gn_params(y, B, P, C, G, gamma, beta, sums);
    p.out = out; p.ldout = C; p.part = part;
    if (ew_gn_fwd(dtype, act, p, ST(stream))) return sgv_set_error(-1, "sgv_op_gn_fwd: launch failed");
    return OPLAUNCH_OK();
}
// weight gradient of the stem convolution, packed layout [N][Kp] fp32: see include/sgvae_ops.h
int sgv_op_stem_conv_dw(const void* x, const void* dy, float* dW, float* part, int B, int H, int W, int N, int KH, int KW, int pad, void* stream) {
    OPCHK(x && dy && dW && part && B > 0 && H > 0 && W > 0 && N > 0, "sgv_op_stem_conv_dw: bad argument");
    OPCHK(KH >= 1 && KH <= 7 && KH == KW && 2 * pad == KH - 1, "sgv_op_stem_conv_dw: square odd windows up to 7x7 with 'same' padding only (got %dx%d pad %d)", KH, KW, pad);
    const int Kp = (KH * KW + 7) / 8 * 8, tx = cdivi(W, STEM_TW), ty = cdivi(H, STEM_TH);
    const int nblk = std::min(B * tx * ty, STEM_DW_BLOCKS);
    for (int n0 = 0; n0 < N; n0 += 32) {
        hipLaunchKernelGGL(stem_conv_dw_kernel, dim3(nblk), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(dy),
                           part, B, H, W, N, n0, pad, tx, ty);
        hipLaunchKernelGGL(stem_dw_finalize_kernel, dim3(32 * Kp), dim3(64), 0, ST(stream), part, dW, nblk, N, n0, KH, KW, Kp);
    }
    return OPLAUNCH_OK();
}
// out = act(gn(y)) with the statistics given (sgv_op_stem_conv_fwd leaves them)
int sgv_op_gn_apply(int dtype, int act, const void* y, void* out, int B, int P, int C, int G, const float* gamma, const float* beta,
                    double* sums, void* stream) {
    OPCHK(y && out && gamma && beta && sums, "sgv_op_gn_apply: null argument");
    OPCHK(C % 8 == 0 && G >= 1 && G <= SGV_GN_MAX_GROUPS && C % G == 0, "sgv_op_gn_apply: C %% 8 == 0, 1 <= G <= %d, C %% G == 0 required", SGV_GN_MAX_GROUPS);
    GNParams p = gn_params(y, B, P, C, G, gamma, beta, sums);
    p.out = out; p.ldout = C;
    if (ew_gn_apply(dtype, act, p, ST(stream))) return sgv_set_error(-1, "sgv_op_gn_apply: launch failed");
    return OPLAUNCH_OK();
}
size_t sgv_op_gn_workspace_floats(int B, int P, int C) { return ew_gn_part_floats(B, P, C); }
// residual-block tail in one pass: see include/sgvae_ops.h
int sgv_op_gn_tail(int dtype, const void* y, const float* gamma, const float* beta, double* sums, const void* y2, const float* gamma2,
                   const float* beta2, double* sums2, const float* cscale, void* out, int B, int P, int C, int G, float* part, void* stream) {
    OPCHK(y && gamma && beta && sums && y2 && out && part, "sgv_op_gn_tail: null argument");
    OPCHK(cscale || (gamma2 && beta2 && sums2), "sgv_op_gn_tail: the second operand needs either cscale or gamma2 / beta2 / sums2");
    OPCHK(C % 8 == 0 && G >= 1 && G <= SGV_GN_MAX_GROUPS && C % G == 0, "sgv_op_gn_tail: C %% 8 == 0, 1 <= G <= %d, C %% G == 0 required", SGV_GN_MAX_GROUPS);
    GNParams p = gn_params(y, B, P, C, G, gamma, beta, sums);
    p.part = part;
    if (ew_gn_stats(dtype, p, ST(stream))) return sgv_set_error(-1, "sgv_op_gn_tail: statistics launch failed");
    GNTail t;
    t.y2 = y2; t.ldy2 = C; t.cscale = cscale;
    if (!cscale) {
        GNParams q = gn_params(y2, B, P, C, G, gamma2, beta2, sums2);
        q.part = part;                  // same stream: the first statistics pass has consumed its partials
        if (ew_gn_stats(dtype, q, ST(stream))) return sgv_set_error(-1, "sgv_op_gn_tail: statistics launch failed");
        t.gamma2 = gamma2; t.beta2 = beta2; t.sums2 = sums2;
    }
    p.out = out; p.ldout = C;
    if (ew_gn_tail(dtype, p, t, ST(stream))) return sgv_set_error(-1, "sgv_op_gn_tail: launch failed");
    return OPLAUNCH_OK();
}
// backward of out = act(gn(y)): dy (same dtype) and dgamma/dbeta (+= , fp32); sums from the forward; sums2: B*G*2 doubles
// scratch; part: sgv_op_gn_workspace_floats floats scratch
static int gn_bwd_impl(int dtype, int act, const void* y, const void* dout, void* dy, int B, int P, int C, int G, const float* gamma,
                       const float* beta, double* sums, double* sums2, float* part, float* dgamma, float* dbeta, int accumulate, void* stream) {
    OPCHK(y && dout && dy && gamma && beta && sums && sums2 && part && dgamma && dbeta, "sgv_op_gn_bwd: null argument");
    OPCHK(C % 8 == 0 && G >= 1 && G <= SGV_GN_MAX_GROUPS && C % G == 0, "sgv_op_gn_bwd: bad channel / group counts");
    GNParams p = gn_params(y, B, P, C, G, gamma, beta, sums);
    p.sums2 = sums2; p.dout = dout; p.lddout = C; p.rscale = 1.f; p.dgamma = dgamma; p.dbeta = dbeta; p.part = part;
    p.out = dy; p.ldout = C; p.accum_affine = accumulate;
    if (ew_gn_bwd(dtype, act, p, ST(stream))) return sgv_set_error(-1, "sgv_op_gn_bwd: launch failed");
    return OPLAUNCH_OK();
}
int sgv_op_gn_bwd(int dtype, int act, const void* y, const void* dout, void* dy, int B, int P, int C, int G, const float* gamma,
                  const float* beta, double* sums, double* sums2, float* part, float* dgamma, float* dbeta, void* stream) {
    return gn_bwd_impl(dtype, act, y, dout, dy, B, P, C, G, gamma, beta, sums, sums2, part, dgamma, dbeta, 1, stream);
}
// the same with dgamma / dbeta written (=) instead of accumulated: no zero-fill needed in front of it
int sgv_op_gn_bwd_set(int dtype, int act, const void* y, const void* dout, void* dy, int B, int P, int C, int G, const float* gamma,
                      const float* beta, double* sums, double* sums2, float* part, float* dgamma, float* dbeta, void* stream) {
    return gn_bwd_impl(dtype, act, y, dout, dy, B, P, C, G, gamma, beta, sums, sums2, part, dgamma, dbeta, 0, stream);
}

// ---- small fp32 layers ------------------------------------------------------------------------------------
int sgv_op_linear_fwd(const float* x, const float* W, const float* bias, const float* scale, float* y, int B, int K, int O, int act, void* stream) {
    OPCHK(x && W && y && B > 0 && K > 0 && O > 0, "sgv_op_linear_fwd: bad argument");
    // (one wave per output handling all batch rows -- every weight row read once -- measured 3x slower at batch 16: 2048 waves
    // with 32 dependent load rounds each cannot hide the L2 latency that 32768 short waves do)
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(O, B), dim3(64), 0, ST(stream), x, W, bias, scale, y, K, O, act);
    return OPLAUNCH_OK();
}
int sgv_op_act_fwd(const float* x, float* y, long n, int act, void* stream) {
    OPCHK(x && y && n > 0, "sgv_op_act_fwd: bad argument");
    hipLaunchKernelGGL(act_fwd_kernel, grid1(n), dim3(256), 0, ST(stream), x, y, n, act);
    return OPLAUNCH_OK();
}
int sgv_op_act_bwd(const float* y, const float* dy, float* dz, long n, int act, void* stream) {
    OPCHK(y && dy && dz && n > 0, "sgv_op_act_bwd: bad argument");
    hipLaunchKernelGGL(act_bwd_kernel, grid1(n), dim3(256), 0, ST(stream), y, dy, dz, n, act);
    return OPLAUNCH_OK();
}
int sgv_op_linear_bwd(const float* dz, const float* x, const float* W, const float* scale, float* dx, int accumulate_dx, float* dW, float* db,
                      int B, int K, int O, void* stream) {
    OPCHK(dz && x && W && (dW || (dx && !db)) && B > 0 && K > 0 && O > 0, "sgv_op_linear_bwd: bad argument");
    if (dx) {
        // enough output chunks for ~1000 blocks, at least 32 outputs each
        const int kb = cdivi(K, 64), bb = cdivi(B, 8);
        int chunks = cdivi(1024, kb * bb);
        if (chunks > cdivi(O, 32)) chunks = cdivi(O, 32);
        if (chunks < 1) chunks = 1;
        const int ochunk = cdivi(cdivi(O, chunks), 4) * 4;
        const int nch = cdivi(O, ochunk);
        float* part = (nch > 1 || accumulate_dx) ? (float*)lc_workspace(ST(stream), sizeof(float) * (size_t)nch * B * K) : dx;
        OPCHK(part, "sgv_op_linear_bwd: workspace allocation failed");
        hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(kb, bb, nch), dim3(256), 0, ST(stream), dz, W, scale, part, K, O, B, ochunk);
        if (part != dx) fin_rows(part, nch, (long)B * K, dx, accumulate_dx != 0, ST(stream));
    }
    if (dW) hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(cdivi(K, 256), cdivi(O, 8)), dim3(256), 0, ST(stream), dz, x, scale, dW, db, B, K, O);
    return OPLAUNCH_OK();
}
int sgv_op_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stat, int B, int K, void* stream) {
    OPCHK(x && gamma && beta && y && stat && B > 0 && K > 0, "sgv_op_layernorm_fwd: bad argument");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(B), dim3(64), 0, ST(stream), x, gamma, beta, y, stat, K);
    return OPLAUNCH_OK();
}
int sgv_op_layernorm_bwd(const float* x, const float* gamma, const float* stat, const float* dy, float* dx, float* dgamma, float* dbeta,
                         int B, int K, void* stream) {
    OPCHK(x && gamma && stat && dy && dx && dgamma && dbeta && B > 0 && K > 0, "sgv_op_layernorm_bwd: bad argument");
    float* part = (float*)lc_workspace(ST(stream), sizeof(float) * 2 * (size_t)B * K);
    OPCHK(part, "sgv_op_layernorm_bwd: workspace allocation failed");
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(B), dim3(64), 0, ST(stream), x, gamma, stat, dy, dx, part, part + (size_t)B * K, K);
    fin_rows(part, B, (long)K, dgamma, false, ST(stream));
    fin_rows(part + (size_t)B * K, B, (long)K, dbeta, false, ST(stream));
    return OPLAUNCH_OK();
}
int sgv_op_batchnorm_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float* y, float* stat,
                         int B, int K, int train, void* stream) {
    OPCHK(x && gamma && beta && run_mean && run_var && y && stat && B > 0 && K > 0, "sgv_op_batchnorm_fwd: bad argument");
    OPCHK(!train || B > 1, "BatchNorm1d in training mode needs more than one value per channel (as torch raises)");
    hipLaunchKernelGGL(batchnorm_fwd_kernel, dim3(cdivi(K, 256)), dim3(256), 0, ST(stream), x, gamma, beta, run_mean, run_var, y, stat, B, K, train);
    return OPLAUNCH_OK();
}
int sgv_op_batchnorm_bwd(const float* x, const float* gamma, const float* stat, const float* dy, float* dx, float* dgamma, float* dbeta,
                         int B, int K, int train, void* stream) {
    OPCHK(x && gamma && stat && dy && dx && dgamma && dbeta && B > 0 && K > 0, "sgv_op_batchnorm_bwd: bad argument");
    hipLaunchKernelGGL(batchnorm_bwd_kernel, dim3(cdivi(K, 256)), dim3(256), 0, ST(stream), x, gamma, stat, dy, dx, dgamma, dbeta, B, K, train);
    return OPLAUNCH_OK();
}
int sgv_op_mask_scale(const float* a, const float* mask, float scale, float* out, long n, void* stream) {
    OPCHK(a && out && n > 0, "sgv_op_mask_scale: bad argument");
    hipLaunchKernelGGL(mask_scale_kernel, grid1(n), dim3(256), 0, ST(stream), a, mask, scale, out, n);
    return OPLAUNCH_OK();
}
int sgv_op_addf(const float* a, const float* b, float* out, long n, void* stream) {
    OPCHK(a && b && out && n > 0, "sgv_op_addf: bad argument");
    hipLaunchKernelGGL(addf_kernel, grid1(n), dim3(256), 0, ST(stream), a, b, out, n);
    return OPLAUNCH_OK();
}
int sgv_op_mse(const float* pred, const float* target, double* loss_dev, float* dpred, float gscale, long n, void* stream) {
    OPCHK(pred && target && loss_dev && n > 0, "sgv_op_mse: bad argument");
    const dim3 g = grid1(n);
    double* part = g.x > 1 ? (double*)lc_workspace(ST(stream), sizeof(double) * g.x) : loss_dev;
    OPCHK(part, "sgv_op_mse: workspace allocation failed");
    hipLaunchKernelGGL(mse_kernel, g, dim3(256), 0, ST(stream), pred, target, part, dpred, gscale, n);
    if (g.x > 1) fin_rows(part, (int)g.x, 1L, loss_dev, false, ST(stream));
    return OPLAUNCH_OK();
}
int sgv_op_loss_value(int kind, const float* a, const float* b, double* loss_dev, float delta, long n, void* stream) {
    OPCHK(a && b && loss_dev && n > 0 && kind >= 0 && kind <= 3, "sgv_op_loss_value: bad argument");
    OPCHK((((uintptr_t)a | (uintptr_t)b) & 15) == 0, "sgv_op_loss_value: operands must be 16-byte aligned");
    OPCHK(kind < 2 || delta > 0.f, "sgv_op_loss_value: delta / beta must be positive");
    long blocks = ((n >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const dim3 g((unsigned)blocks);
    double* part = blocks > 1 ? (double*)lc_workspace(ST(stream), sizeof(double) * blocks) : loss_dev;
    OPCHK(part, "sgv_op_loss_value: workspace allocation failed");
    if (kind == 0) hipLaunchKernelGGL(loss_value_kernel<0>, g, dim3(256), 0, ST(stream), a, b, part, delta, n);
    else if (kind == 1) hipLaunchKernelGGL(loss_value_kernel<1>, g, dim3(256), 0, ST(stream), a, b, part, delta, n);
    else if (kind == 2) hipLaunchKernelGGL(loss_value_kernel<2>, g, dim3(256), 0, ST(stream), a, b, part, delta, n);
    else hipLaunchKernelGGL(loss_value_kernel<3>, g, dim3(256), 0, ST(stream), a, b, part, delta, n);
    if (blocks > 1) fin_rows(part, (int)blocks, 1L, loss_dev, false, ST(stream));
    return OPLAUNCH_OK();
}
int sgv_op_cols_sub_div(const float* x, const float* col_min, const float* col_scale, float* y, long rows, int cols, void* stream) {
    OPCHK(x && col_min && col_scale && y && rows > 0 && cols > 0, "sgv_op_cols_sub_div: bad argument");
    hipLaunchKernelGGL(cols_sub_div_kernel, grid1(rows * cols), dim3(256), 0, ST(stream), x, col_min, col_scale, y, rows, cols);
    return OPLAUNCH_OK();
}
// C[M][N] = scale * A[M][K] . W[N][K]^T (+ bias[N]) (+ addend[M][N]); K, N multiples of 8; out_f32: fp32 output
int sgv_op_gemm_nt(int dtype, const void* A, const void* W, void* C, const float* bias, const float* scale, const void* addend,
                   int M, int N, int K, int out_f32, void* stream) {
    OPCHK(A && W && C && M > 0 && N > 0 && K > 0, "sgv_op_gemm_nt: bad argument");
    OPCHK(K % 8 == 0 && N % 8 == 0, "sgv_op_gemm_nt: K and N must be multiples of 8 (got K=%d N=%d)", K, N);
    GemmNT p; memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.w_tap_stride = (long)N * K; p.C = C; p.ldc = N;
    p.addend = addend; p.ldadd = N; p.bias = bias; p.scale = scale;
    p.M = M; p.N = N; p.K = K; p.taps = 1; p.pad = 0; p.Tlen = M; p.splitk = 1; p.out_f32 = out_f32;
    // the engine's kernel choice (gemm256.hip): 256x256 persistent kernel for the big products, 128-row kernels otherwise;
    // this stateless entry point has no split-K workspace, so every plan is split-K 1
    const GemmPlan pl = gemm_nt_plan(dtype, p, 0, 0);
    const int r = launch_gemm_nt_planned(dtype, p, pl, ST(stream));
    if (r) return sgv_set_error(-1, "sgv_op_gemm_nt: launch rejected (%d) for M=%d N=%d K=%d", r, M, N, K);
    return 0;
}
// Implicit-GEMM 2-D convolution on a channels-last batch (no im2col matrix): see include/sgvae_ops.h.
int sgv_op_conv2d_nt(int dtype, const void* x, const void* W, void* y, const float* scale, int B, int H, int Wd, int Cin, int N,
                     int KH, int KW, int stride, int pad, long ldw, long w_tap_stride, int flip, void* stream) {
    OPCHK(x && W && y && B > 0 && H > 0 && Wd > 0 && Cin > 0 && N > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
          "sgv_op_conv2d_nt: bad argument");
    OPCHK(Cin % 8 == 0 && N % 8 == 0 && ldw % 8 == 0 && w_tap_stride % 8 == 0,
          "sgv_op_conv2d_nt: channels, ldw and the tap stride must be multiples of 8 (got Cin=%d N=%d ldw=%ld tap stride=%ld)", Cin, N, ldw, w_tap_stride);
    OPCHK(KH * KW <= 31, "sgv_op_conv2d_nt: at most 31 taps (got %dx%d)", KH, KW);
    OPCHK(H + 2 * pad >= KH && Wd + 2 * pad >= KW, "sgv_op_conv2d_nt: window larger than the padded image");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    OPCHK((long)B * Ho * Wo < 0x7FFFFFFFL, "sgv_op_conv2d_nt: too many output pixels");
    GemmNT p; memset(&p, 0, sizeof(p));
    p.A = x; p.lda = Cin; p.W = W; p.ldw = ldw; p.w_tap_stride = w_tap_stride; p.C = y; p.ldc = N; p.scale = scale;
    p.M = B * Ho * Wo; p.N = N; p.K = Cin; p.taps = KH * KW; p.pad = 0; p.Tlen = p.M; p.splitk = 1;
    p.a_rows = (long)B * H * Wd;
    p.cv_kw = KW; p.cv_H = H; p.cv_W = Wd; p.cv_S = stride; p.cv_P = pad; p.cv_Ho = Ho; p.cv_Wo = Wo; p.cv_flip = flip ? 1 : 0;
    const GemmPlan pl = gemm_nt_plan(dtype, p, 0, 0);
    const int r = launch_gemm_nt_planned(dtype, p, pl, ST(stream));
    if (r) return sgv_set_error(-1, "sgv_op_conv2d_nt: launch rejected (%d) for B=%d H=%d W=%d Cin=%d N=%d %dx%d/%d", r, B, H, Wd, Cin, N, KH, KW, stride);
    return 0;
}
// C = scale * A W^T + up2(addend): see include/sgvae_ops.h
int sgv_op_gemm_nt_add_s2(int dtype, const void* A, const void* W, void* C, const float* scale, const void* addend, int M, int N, int K,
                          int H, int Wd, void* stream) {
    OPCHK(A && W && C && addend && M > 0 && N > 0 && K > 0 && H > 0 && Wd > 0, "sgv_op_gemm_nt_add_s2: bad argument");
    OPCHK(dtype == 1, "sgv_op_gemm_nt_add_s2: bf16 only");
    OPCHK(K % 8 == 0 && N % 8 == 0 && M % (H * Wd) == 0, "sgv_op_gemm_nt_add_s2: K, N multiples of 8 and M a multiple of H*W required (got M=%d N=%d K=%d H=%d W=%d)", M, N, K, H, Wd);
    GemmNT p; memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.w_tap_stride = (long)N * K; p.C = C; p.ldc = N;
    p.addend = addend; p.ldadd = N; p.scale = scale; p.add_H = H; p.add_W = Wd;
    p.M = M; p.N = N; p.K = K; p.taps = 1; p.pad = 0; p.Tlen = M; p.splitk = 1;
    const GemmPlan pl = gemm_nt_plan(dtype, p, 0, 0);
    const int r = launch_gemm_nt_planned(dtype, p, pl, ST(stream));
    if (r) return sgv_set_error(-1, "sgv_op_gemm_nt_add_s2: launch rejected (%d) for M=%d N=%d K=%d (K >= 2048 with N >= 256 is not served)", r, M, N, K);
    return 0;
}
// dW[N1][N2] (fp32) = A[M][N1]^T . B[M][N2]; N1, N2 multiples of 8.  The reduction runs over the M = B*H*W rows, up to
// a million of them for a handful of output tiles: sgv_op_gemm_tn_splitk() says how many row slices to use and the caller
// provides splitk * N1 * N2 floats of slab workspace (deterministic: plain stores + one sum pass).
int sgv_op_gemm_tn_splitk(int dtype, int M, int N1, int N2) {
    const long steps = cdivi(M, dtype == 1 ? 32 : 16);
    long tiles = (long)cdivi(N1, 128) * cdivi(N2, 128), slots = 768;      // three 128x128 blocks per CU
    if (gemm_tn_uses_w2(dtype, M, N1, N2, M)) { tiles = (long)cdivi(N1, 128) * cdivi(N2, 256); slots = 512; }   // two 128x256 blocks per CU
    long sk = slots / tiles;                     // one full round of blocks
    if (sk > steps / 8) sk = steps / 8;          // keep >= 8 K-steps per slice
    if (sk > 256) sk = 256;
    return sk < 1 ? 1 : (int)sk;
}
// out[i] = sum_z slabs[z][i]; the conditioner's weight gradients are small matrices cut into up to 256 slabs, so a block
// takes 64 elements x 4 slab lanes (4 independent chains per element instead of one long one)
__global__ __launch_bounds__(256) void op_sum_slabs_kernel(float* out, const float* slabs, int splitk, long n) {
    __shared__ float sm[4][64];
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    for (long i0 = (long)blockIdx.x * 64; i0 < n; i0 += (long)gridDim.x * 64) {
        const long i = i0 + el;
        float v = 0.f;
        if (i < n)
            for (int z = zl; z < splitk; z += 4) v += slabs[(long)z * n + i];
        sm[zl][el] = v;
        __syncthreads();
        if (zl == 0 && i < n) out[i] = sm[0][el] + sm[1][el] + sm[2][el] + sm[3][el];
        __syncthreads();
    }
}
int sgv_op_gemm_tn(int dtype, const void* A, const void* Bm, float* dW, int M, int N1, int N2, float* slabs, int splitk, void* stream) {
    OPCHK(A && Bm && dW && M > 0 && N1 > 0 && N2 > 0, "sgv_op_gemm_tn: bad argument");
    OPCHK(N1 % 8 == 0 && N2 % 8 == 0, "sgv_op_gemm_tn: N1 and N2 must be multiples of 8 (got %d, %d)", N1, N2);
    OPCHK(splitk <= 1 || slabs, "sgv_op_gemm_tn: split-K needs a slab workspace");
    GemmTN p; memset(&p, 0, sizeof(p));
    p.A = A; p.lda = N1; p.B = Bm; p.ldb = N2; p.ldo = N2; p.out_tap_stride = (long)N1 * N2;
    p.M = M; p.N1 = N1; p.N2 = N2; p.taps = 1; p.pad = 0; p.Tlen = M; p.use_tr = 1;
    p.splitk = splitk < 1 ? 1 : splitk;
    p.out = p.splitk > 1 ? slabs : dW; p.out_slab_stride = (long)N1 * N2;
    const int r = launch_gemm_tn(dtype, p, ST(stream));
    if (r) return sgv_set_error(-1, "sgv_op_gemm_tn: launch rejected (%d) for M=%d N1=%d N2=%d", r, M, N1, N2);
    if (p.splitk > 1)
        hipLaunchKernelGGL(op_sum_slabs_kernel, grid1((long)N1 * N2 * 4), dim3(256), 0, ST(stream), dW, slabs, p.splitk, (long)N1 * N2);
    return OPLAUNCH_OK();
}
// Weight gradient of a 2-D convolution without the im2col matrix: see include/sgvae_ops.h.
int sgv_op_conv2d_tn(int dtype, const void* dy, const void* x, float* dW, int B, int H, int Wd, int Cin, int N1, int KH, int KW,
                     int stride, int pad, float* slabs, int splitk, void* stream) {
    OPCHK(dy && x && dW && B > 0 && H > 0 && Wd > 0 && Cin > 0 && N1 > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
          "sgv_op_conv2d_tn: bad argument");
    OPCHK(Cin % 8 == 0 && N1 % 8 == 0, "sgv_op_conv2d_tn: channels must be multiples of 8 (got Cin=%d Cout=%d)", Cin, N1);
    OPCHK(H + 2 * pad >= KH && Wd + 2 * pad >= KW, "sgv_op_conv2d_tn: window larger than the padded image");
    OPCHK(splitk <= 1 || slabs, "sgv_op_conv2d_tn: split-K needs a slab workspace");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    OPCHK((long)B * Ho * Wo < 0x7FFFFFFFL, "sgv_op_conv2d_tn: too many output pixels");
    const int N2 = KH * KW * Cin;
    GemmTN p; memset(&p, 0, sizeof(p));
    p.A = dy; p.lda = N1; p.B = x; p.ldb = Cin; p.ldo = N2; p.out_tap_stride = (long)N1 * N2;
    p.M = B * Ho * Wo; p.N1 = N1; p.N2 = N2; p.taps = 1; p.pad = 0; p.Tlen = p.M; p.use_tr = 1;
    p.splitk = splitk < 1 ? 1 : splitk;
    p.out = p.splitk > 1 ? slabs : dW; p.out_slab_stride = (long)N1 * N2;
    p.cv_kw = KW; p.cv_H = H; p.cv_W = Wd; p.cv_S = stride; p.cv_P = pad; p.cv_Ho = Ho; p.cv_Wo = Wo; p.cv_C = Cin;
    const int r = launch_gemm_tn(dtype, p, ST(stream));
    if (r) return sgv_set_error(-1, "sgv_op_conv2d_tn: launch rejected (%d) for B=%d H=%d W=%d Cin=%d Cout=%d %dx%d/%d", r, B, H, Wd, Cin, N1, KH, KW, stride);
    if (p.splitk > 1)
        hipLaunchKernelGGL(op_sum_slabs_kernel, grid1((long)N1 * N2 * 4), dim3(256), 0, ST(stream), dW, slabs, p.splitk, (long)N1 * N2);
    return OPLAUNCH_OK();
}
// dst_k[0..count_k) = src_k[0..count_k) for every row k of a device table {src, dst, count} (fp32 tensors): the host model
// files ~110 freshly computed gradients into its flat arena with one launch instead of one copy each
struct CopyRow { const float* src; float* dst; long count; };
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyRow* rows) {
    const CopyRow r = rows[blockIdx.y];
    const bool vec = ((((uintptr_t)r.src) | ((uintptr_t)r.dst)) & 15) == 0;
    const long n4 = vec ? r.count >> 2 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        reinterpret_cast<float4*>(r.dst)[i] = reinterpret_cast<const float4*>(r.src)[i];
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < r.count; i += (long)gridDim.x * 256) r.dst[i] = r.src[i];
}
int sgv_op_multi_copy(const void* table_dev, int n_rows, void* stream) {
    OPCHK(table_dev && n_rows > 0, "sgv_op_multi_copy: bad argument");
    hipLaunchKernelGGL(multi_copy_kernel, dim3(64, n_rows), dim3(256), 0, ST(stream), reinterpret_cast<const CopyRow*>(table_dev));
    return OPLAUNCH_OK();
}
// ---- input augmentation ------------------------------------------------------------------------------------
int sgv_op_flip_roll(const float* x, float* out, int B, int H, int W, const int* flip, const int* shift_x, const int* shift_y, void* stream) {
    OPCHK(x && out && flip && shift_x && shift_y && B > 0 && H > 0 && W > 0, "sgv_op_flip_roll: bad argument");
    hipLaunchKernelGGL(flip_roll_kernel, grid1((long)B * H * W), dim3(256), 0, ST(stream), x, out, B, H, W, flip, shift_x, shift_y);
    return OPLAUNCH_OK();
}
int sgv_op_affine_sample(const float* x, float* out, int B, int H, int W, const float* theta, void* stream) {
    OPCHK(x && out && theta && B > 0 && H > 0 && W > 0, "sgv_op_affine_sample: bad argument");
    hipLaunchKernelGGL(affine_sample_kernel, grid1((long)B * H * W), dim3(256), 0, ST(stream), x, out, B, H, W, theta);
    return OPLAUNCH_OK();
}
int sgv_op_mixup_rows(const float* x, const int* perm, float lam, float* out, int B, long n, void* stream) {
    OPCHK(x && perm && out && B > 0 && n > 0, "sgv_op_mixup_rows: bad argument");
    hipLaunchKernelGGL(mixup_rows_kernel, grid1((long)B * n), dim3(256), 0, ST(stream), x, perm, lam, out, B, n);
    return OPLAUNCH_OK();
}
// ---- parameter-side helpers --------------------------------------------------------------------------------
int sgv_op_l2_normalize(const float* x, float* out, long n, float eps, void* stream) {
    OPCHK(x && out && n > 0, "sgv_op_l2_normalize: bad argument");
    hipLaunchKernelGGL(l2_normalize_kernel, dim3(1), dim3(256), 0, ST(stream), x, out, n, eps);
    return OPLAUNCH_OK();
}
int sgv_op_dot(const float* a, const float* b, float* out4, long n, void* stream) {
    OPCHK(a && b && out4 && n > 0, "sgv_op_dot: bad argument");
    OPCHK(((uintptr_t)out4 & 7) == 0, "sgv_op_dot: out4 must be 8-byte aligned");
    long blocks = (n + 8191) / 8192; if (blocks > 512) blocks = 512; if (blocks < 1) blocks = 1;
    double* part = blocks > 1 ? (double*)lc_workspace(ST(stream), sizeof(double) * blocks) : reinterpret_cast<double*>(out4 + 2);
    OPCHK(part, "sgv_op_dot: workspace allocation failed");
    hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), a, b, part, n);
    if (blocks > 1) fin_rows(part, (int)blocks, 1L, reinterpret_cast<double*>(out4 + 2), false, ST(stream));
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(1), 0, ST(stream), out4);
    return OPLAUNCH_OK();
}
int sgv_op_matvec_t(const float* W, const float* x, float* out, int rows, int cols, void* stream) {
    OPCHK(W && x && out && rows > 0 && cols > 0, "sgv_op_matvec_t: bad argument");
    const int rb = cdivi(rows, 64);
    float* part = rb > 1 ? (float*)lc_workspace(ST(stream), sizeof(float) * (size_t)rb * cols) : out;
    OPCHK(part, "sgv_op_matvec_t: workspace allocation failed");
    hipLaunchKernelGGL(matvec_t_kernel, dim3(cdivi(cols, 256), rb), dim3(256), 0, ST(stream), W, x, part, rows, cols);
    if (rb > 1) fin_rows(part, rb, (long)cols, out, false, ST(stream));
    return OPLAUNCH_OK();
}
int sgv_op_sn_grad(const float* G, const float* u, const float* v, const float* gw, const float* sigma2, float* out, int rows, int cols,
                   void* stream) {
    OPCHK(G && u && v && gw && sigma2 && out && rows > 0 && cols > 0, "sgv_op_sn_grad: bad argument");
    hipLaunchKernelGGL(sn_grad_kernel, grid1((long)rows * cols), dim3(256), 0, ST(stream), G, u, v, gw, sigma2, out, rows, cols);
    return OPLAUNCH_OK();
}
int sgv_op_conv_weight_pack(int dtype, const float* w, void* packed, int Cout, int Cin, int KH, int KW, void* stream) {
    OPCHK(w && packed && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "sgv_op_conv_weight_pack: bad argument");
    const int Kp = (KH * KW * Cin + 7) / 8 * 8;
    ON_DTYPE(dtype, hipLaunchKernelGGL(conv_weight_pack_kernel<T>, grid1((long)Cout * Kp), dim3(256), 0, ST(stream), w, PT(packed), Cout, Cin, KH, KW, Kp));
    return OPLAUNCH_OK();
}
int sgv_op_conv_weight_unpack(const float* packed, float* w, int Cout, int Cin, int KH, int KW, void* stream) {
    OPCHK(w && packed && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "sgv_op_conv_weight_unpack: bad argument");
    const int Kp = (KH * KW * Cin + 7) / 8 * 8;
    hipLaunchKernelGGL(conv_weight_unpack_kernel, grid1((long)Cout * Cin * KH * KW), dim3(256), 0, ST(stream), packed, w, Cout, Cin, KH, KW, Kp);
    return OPLAUNCH_OK();
}
int sgv_op_sumsq(const float* g, long n, double* acc, void* stream) {
    OPCHK(g && acc && n > 0, "sgv_op_sumsq: bad argument");
    long blocks = (n + 4095) / 4096; if (blocks > 256) blocks = 256; if (blocks < 1) blocks = 1;
    double* part = (double*)lc_workspace(ST(stream), sizeof(double) * blocks);
    OPCHK(part, "sgv_op_sumsq: workspace allocation failed");
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), g, n, part);
    fin_rows(part, (int)blocks, 1L, acc, true, ST(stream));
    return OPLAUNCH_OK();
}
int sgv_op_clip_coef(const double* sumsq, float max_norm, float* out2, void* stream) {
    OPCHK(sumsq && out2, "sgv_op_clip_coef: bad argument");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, ST(stream), sumsq, max_norm, out2);
    return OPLAUNCH_OK();
}
int sgv_op_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                 int step, const float* gscale, void* stream) {
    OPCHK(p && g && m && v && n > 0 && step >= 1, "sgv_op_adamw: bad argument");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step)), bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adamw_flat_kernel, grid1(n), dim3(256), 0, ST(stream), p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, gscale);
    return OPLAUNCH_OK();
}
// [Bn][I][J] -> [Bn][J][I] with dtype conversion (NCHW fp32 <-> channels-last compute dtype)
int sgv_op_transpose(int src_dtype, int dst_dtype, const void* src, void* dst, int Bn, int I, int J, void* stream) {
    OPCHK(src && dst && Bn > 0 && I > 0 && J > 0, "sgv_op_transpose: bad argument");
    ew_transpose(src_dtype, dst_dtype, src, dst, Bn, I, J, J, I, (long)I * J, (long)I * J, ST(stream));
    return OPLAUNCH_OK();
}

}  // extern "C"
