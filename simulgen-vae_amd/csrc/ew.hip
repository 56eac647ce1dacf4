// HBM-bound kernels of the SimulGen-VAE step on channels-last [B*T][C] maps (gfx950):
// GroupNorm(+GELU/tanh) forward/backward (reference nn.GroupNorm/nn.GELU call sites
// modules/encoder.py:35-36, modules/common.py:85-87,111-112,136-143, modules/decoder.py:119-120,
// 136-137,146-147), the reconstruction loss fused into the recon-head GroupNorm pass
// (modules/VAE_network.py:110-111), reparameterisation + KL terms (modules/decoder.py:199-223,
// modules/losses.py:8-48), the small Linear heads (modules/encoder.py:140-142,156-162,
// modules/decoder.py:133,143), layout conversion and augmentation (modules/augmentation.py:86-124).
// All arithmetic in fp32 (group sums in fp64); every thread owns 8 consecutive channels (16/32-byte
// vector accesses) and walks rows, so per-channel constants stay in registers.
#include <algorithm>
#include "sgv_ew.h"

static inline int cdiv_i(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// geometry shared by the GroupNorm-family kernels
// ------------------------------------------------------------------------------------------
struct GNGeom {
    int CV, RL, rowsplit;
    dim3 grid;
};
static GNGeom gn_geom(int B, int T, int C, int target = 2048) {
    GNGeom g;
    const int nv = C / 8;
    int cv = 1;
    while (cv < nv && cv < 256) cv <<= 1;
    g.CV = cv;
    g.RL = 256 / cv;
    const int colblocks = cdiv_i(nv, cv);
    // aim for >= ~target blocks, at least RL rows per block
    int rs = cdiv_i(target, (long)colblocks * B);
    int maxrs = cdiv_i(T, g.RL);
    if (rs > maxrs) rs = maxrs;
    if (rs < 1) rs = 1;
    g.rowsplit = rs;
    g.grid = dim3(colblocks, rs, B);
    return g;
}

struct GNCtx {
    int tx, ty, c0, b, t_lo, t_hi, RL;
    bool col_ok;
};
__device__ __forceinline__ GNCtx gn_ctx(const GNParams& p) {
    GNCtx c;
    const int tid = threadIdx.x;
    c.tx = tid % p.CV;
    c.ty = tid / p.CV;
    c.RL = 256 / p.CV;
    c.c0 = (blockIdx.x * p.CV + c.tx) * 8;
    c.col_ok = c.c0 < p.C;
    c.b = blockIdx.z;
    const int rps = (p.T + gridDim.y - 1) / gridDim.y;
    c.t_lo = blockIdx.y * rps;
    c.t_hi = min(p.T, c.t_lo + rps);
    return c;
}

// per-element mean / rstd (and, WITH_M, the backward means m1 = s1/n, m2 = s2/n) of the element's group.
// The fp64 divide/sqrt chain runs once per block on G threads and is broadcast through LDS: done per
// thread it cost more than the streaming work of the small layers.  Must be called by all 256 threads.
template <bool WITH_M = false>
__device__ __forceinline__ void gn_consts(const GNParams& p, const GNCtx& c, float mean[8], float rstd[8],
                                          float* m1 = nullptr, float* m2 = nullptr) {
    __shared__ float gc[SGV_GN_MAX_GROUPS][4];
    if ((int)threadIdx.x < p.G) {
        const int g = threadIdx.x;
        const double n = (double)p.Cg * (double)p.T;
        const double s = p.sums[((long)c.b * p.G + g) * 2 + 0];
        const double ss = p.sums[((long)c.b * p.G + g) * 2 + 1];
        const double m = s / n;
        double var = ss / n - m * m;
        if (var < 0.0) var = 0.0;
        gc[g][0] = (float)m;
        gc[g][1] = (float)(1.0 / sqrt(var + 1e-5));
        if constexpr (WITH_M) {
            gc[g][2] = (float)(p.sums2[((long)c.b * p.G + g) * 2 + 0] / n);
            gc[g][3] = (float)(p.sums2[((long)c.b * p.G + g) * 2 + 1] / n);
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = min((c.c0 + e) / p.Cg, p.G - 1);
        mean[e] = gc[g][0];
        rstd[e] = gc[g][1];
        if constexpr (WITH_M) { m1[e] = gc[g][2]; m2[e] = gc[g][3]; }
    }
}

// sum of one float per thread over the block in a fixed order (shuffle tree per wave, then the waves in index order);
// the result is returned to every thread.  All blockDim.x (<= 1024, multiple of 64) threads must call.
__device__ __forceinline__ float block_sum_fixed(float v, float* smw) {
    const float w = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smw[threadIdx.x >> 6] = w;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += smw[i];
    return t;
}
// sum of one double per thread over the block (<= 1024 threads, multiple of 64) in a fixed order, result broadcast to every
// thread (all threads call); sm16: 16 doubles of LDS
__device__ __forceinline__ double block_sum_f64(double v, double* sm16) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm16[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm16[i];
    return t;
}
// linear index of this block in its grid
__device__ __forceinline__ int block_linear() { return ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x; }

// Block-level reduction for the statistics pass: column sums over the block's rows, then per-group sums of this block's
// columns, written (not accumulated) to gpart[block_linear()][G][2]; ew_gn_stats sums the blocks of a sample in a fixed order.
__device__ __forceinline__ void gn_block_group_partials(const GNParams& p, const GNCtx& c, float colA[8], float colB[8], float* gpart) {
    __shared__ float smA[2048];
    __shared__ float smB[2048];
    __shared__ float smw[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        smA[(c.ty * p.CV + c.tx) * 8 + e] = colA[e];
        smB[(c.ty * p.CV + c.tx) * 8 + e] = colB[e];
    }
    __syncthreads();
    const bool owner = c.ty == 0 && c.col_ok;
    if (owner) {
        for (int r = 1; r < c.RL; ++r) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                colA[e] += smA[(r * p.CV + c.tx) * 8 + e];
                colB[e] += smB[(r * p.CV + c.tx) * 8 + e];
            }
        }
    }
    float* dst = gpart + (long)block_linear() * p.G * 2;
    const int blk_lo = blockIdx.x * p.CV * 8, blk_hi = (int)min((long)(blockIdx.x + 1) * p.CV * 8, (long)p.C);
    const int g_lo = min(blk_lo / p.Cg, p.G - 1);
    const int g_hi = min((blk_hi - 1) / p.Cg, p.G - 1);
    if (g_hi - g_lo > 3 && p.Cg <= 64) {
        // many narrow groups in one block (the image conditioner: 32 groups of 1-32 channels): the owners' column sums go back
        // to LDS in channel order and thread g adds the channels of group g in index order -- one barrier instead of two
        // block-wide reductions per group
        __syncthreads();
        if (owner) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { smA[c.tx * 8 + e] = colA[e]; smB[c.tx * 8 + e] = colB[e]; }
        }
        __syncthreads();
        for (int g = threadIdx.x; g < p.G; g += 256) {
            float ga = 0.f, gb = 0.f;
            if (g >= g_lo && g <= g_hi) {
                const int lo = max(g * p.Cg, blk_lo), hi = min(g == p.G - 1 ? p.C : (g + 1) * p.Cg, blk_hi);
                for (int ch = lo; ch < hi; ++ch) { ga += smA[ch - blk_lo]; gb += smB[ch - blk_lo]; }
            }
            dst[g * 2] = ga; dst[g * 2 + 1] = gb;
        }
        return;
    }
    for (int g = 0; g < p.G; ++g) {
        float ga = 0.f, gb = 0.f;
        if (g >= g_lo && g <= g_hi) {            // block-uniform
            float a = 0.f, b2 = 0.f;
            if (owner) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (min((c.c0 + e) / p.Cg, p.G - 1) == g) { a += colA[e]; b2 += colB[e]; }
            }
            ga = block_sum_fixed(a, smw);
            gb = block_sum_fixed(b2, smw);
        }
        if (threadIdx.x == 0) { dst[g * 2] = ga; dst[g * 2 + 1] = gb; }
    }
}

// Contention-free alternative for the backward passes: every block writes its NARR column-sum arrays
// (summed over the block's rows) to part[((b*RS + rowblock)*NARR + k)*C + c]; gn_bwd_finalize / colsum
// finalize kernels combine them.  (Per-channel float atomics from ~700 blocks per address ran 10x over
// the bandwidth bound.)
template <int NARR>
__device__ __forceinline__ void gn_block_colsums(const GNParams& p, const GNCtx& c, float (&col)[NARR][8]) {
    __shared__ float sm[NARR][2048];
    if (c.RL > 1) {
#pragma unroll
        for (int k = 0; k < NARR; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) sm[k][(c.ty * p.CV + c.tx) * 8 + e] = col[k][e];
        __syncthreads();
    }
    if (c.ty == 0 && c.col_ok) {
        for (int r = 1; r < c.RL; ++r)
#pragma unroll
            for (int k = 0; k < NARR; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) col[k][e] += sm[k][(r * p.CV + c.tx) * 8 + e];
        float* dst = p.part + ((long)(c.b * gridDim.y + blockIdx.y) * NARR) * p.C + c.c0;
#pragma unroll
        for (int k = 0; k < NARR; ++k) store8(dst + (long)k * p.C, col[k]);
    }
}

// block sum of one float per thread -> dst[block_linear()] (all threads must call); the partials are summed later in a fixed order
__device__ __forceinline__ void block_store_partial(float v, float* dst) {
    __shared__ float smr[16];
    const float t = block_sum_fixed(v, smr);
    if (threadIdx.x == 0) dst[block_linear()] = t;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GNParams p) {
    const GNCtx c = gn_ctx(p);
    float a[8], s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; s[e] = 0.f; }
    if (c.col_ok) {
        const T* y = reinterpret_cast<const T*>(p.y);
        // two rows per round, both loads issued before the first use (streaming pass: latency, not VALU, limits it)
        for (int t = c.t_lo + c.ty; t < c.t_hi; t += 2 * c.RL) {
            const bool two = t + c.RL < c.t_hi;
            Raw8<T> r0, r1;
            raw_load(y + ((long)c.b * p.T + t) * p.ldy + c.c0, r0);
            raw_load(y + ((long)c.b * p.T + (two ? t + c.RL : t)) * p.ldy + c.c0, r1);
            float v[8];
            raw_unpack(r0, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] += v[e]; s[e] += v[e] * v[e]; }
            if (two) {
                raw_unpack(r1, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { a[e] += v[e]; s[e] += v[e] * v[e]; }
            }
        }
    }
    gn_block_group_partials(p, c, a, s, p.part);
}

__device__ __forceinline__ float act_apply(int act, float z) {
    return act == 1 ? gelu_f(z) : (act == 2 ? tanh_f(z) : (act == 3 ? fmaxf(z, 0.f) : z));
}

// out = [res + rscale *] act(gn(y))
template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GNParams p) {
    const GNCtx c = gn_ctx(p);
    float mean[8], rstd[8], ka[8], kb[8];
    gn_consts(p, c, mean, rstd);
    if (!c.col_ok) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float g = p.gamma[c.c0 + e];
        ka[e] = rstd[e] * g;
        kb[e] = p.beta[c.c0 + e] - mean[e] * rstd[e] * g;
    }
    const T* y = reinterpret_cast<const T*>(p.y);
    const T* res = reinterpret_cast<const T*>(p.res);
    T* out = reinterpret_cast<T*>(p.out);
    for (int t = c.t_lo + c.ty; t < c.t_hi; t += 2 * c.RL) {      // two rows per round, loads first
        const bool two = t + c.RL < c.t_hi;
        const long m0 = (long)c.b * p.T + t, m1 = two ? m0 + c.RL : m0;
        Raw8<T> ry0, ry1, rr0, rr1;
        raw_load(y + m0 * p.ldy + c.c0, ry0);
        raw_load(y + m1 * p.ldy + c.c0, ry1);
        if (res) { raw_load(res + m0 * p.ldres + c.c0, rr0); raw_load(res + m1 * p.ldres + c.c0, rr1); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            float v[8], r[8];
            raw_unpack(u ? ry1 : ry0, v);
            if (res) raw_unpack(u ? rr1 : rr0, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = act_apply(ACT, v[e] * ka[e] + kb[e]);
                v[e] = res ? r[e] + p.rscale * f : f;
            }
            store8(out + (u ? m1 : m0) * p.ldout + c.c0, v);
        }
    }
}

// Tail of a residual block of the latent conditioner in one pass (modules/latent_conditioner_model_cnn.py:ResidualBlock.forward):
//   out = relu(A + gn(y))   with A = gn2(y2) (its own statistics and affine; SE = false)
//                                or y2 * cscale[b][c] (y2 already normalised, squeeze-excite scaling; SE = true)
// instead of two normalise passes (or a scale pass) and an add + relu pass over the same rows.
template <typename T, bool SE>
__global__ __launch_bounds__(256) void gn_tail_kernel(const GNParams p, const GNTail t) {
    const GNCtx c = gn_ctx(p);
    float mean[8], rstd[8], ka[8], kb[8], ka2[8], kb2[8];
    gn_consts(p, c, mean, rstd);
    if (c.col_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = p.gamma[c.c0 + e];
            ka[e] = rstd[e] * g;
            kb[e] = p.beta[c.c0 + e] - mean[e] * rstd[e] * g;
        }
    }
    if constexpr (SE) {
        if (c.col_ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { ka2[e] = t.cscale[(long)c.b * p.C + c.c0 + e]; kb2[e] = 0.f; }
        }
    } else {
        __syncthreads();                      // gn_consts broadcasts through one LDS table: everyone has read the first set
        GNParams q = p;
        q.sums = t.sums2;
        gn_consts(q, c, mean, rstd);
        if (c.col_ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = t.gamma2[c.c0 + e];
                ka2[e] = rstd[e] * g;
                kb2[e] = t.beta2[c.c0 + e] - mean[e] * rstd[e] * g;
            }
        }
    }
    if (!c.col_ok) return;
    const T* y = reinterpret_cast<const T*>(p.y);
    const T* y2 = reinterpret_cast<const T*>(t.y2);
    T* out = reinterpret_cast<T*>(p.out);
    for (int r = c.t_lo + c.ty; r < c.t_hi; r += 2 * c.RL) {      // two rows per round, loads first
        const bool two = r + c.RL < c.t_hi;
        const long m0 = (long)c.b * p.T + r, m1 = two ? m0 + c.RL : m0;
        Raw8<T> ry0, ry1, rz0, rz1;
        raw_load(y + m0 * p.ldy + c.c0, ry0);
        raw_load(y + m1 * p.ldy + c.c0, ry1);
        raw_load(y2 + m0 * t.ldy2 + c.c0, rz0);
        raw_load(y2 + m1 * t.ldy2 + c.c0, rz1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            float v[8], z[8];
            raw_unpack(u ? ry1 : ry0, v);
            raw_unpack(u ? rz1 : rz0, z);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // each term is rounded to the compute dtype first: the same values as the separate normalise / scale passes
                const float a = to_f32(from_f32<T>(z[e] * ka2[e] + kb2[e])), b = to_f32(from_f32<T>(v[e] * ka[e] + kb[e]));
                v[e] = fmaxf(a + b, 0.f);
            }
            store8(out + (u ? m1 : m0) * p.ldout + c.c0, v);
        }
    }
}

// d(loss)/d(xhat) for the selected reconstruction loss, unit weight (mean reduction folded by caller)
__device__ __forceinline__ float loss_grad(int lt, float d) {
    if (lt == 0) return 2.f * d;
    if (lt == 1) return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
}
__device__ __forceinline__ float loss_val(int lt, float d) {
    if (lt == 0) return d * d;
    if (lt == 1) return fabsf(d);
    const float a = fabsf(d);
    return a < 1.f ? 0.5f * d * d : a - 0.5f;
}

// Backward reduction pass.  dz = dOut * act'(z) with dOut either a stored gradient (times p.rscale)
// or, FROM_LOSS, the unit-weight loss gradient of tanh(z) vs the target p.dout (= input x).
//   column sums  A_c = sum dz, B_c = sum dz*xhat  -> dbeta, dgamma (per channel)
//   group sums   s1 = sum gamma*dz, s2 = sum gamma*dz*xhat -> p.sums2
// FROM_LOSS also accumulates the loss sums (p.loss_sums[0] selected, [1] squared error) and can
// write xhat.
// LT: the loss kind as a compile-time constant (FROM_LOSS, bf16 engines: without it the selection is four scalar
// branches per element), or -1 = read p.loss_type.
template <typename T, int ACT, bool FROM_LOSS, bool TRAIN, int LT = -1>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const GNParams p) {
    const GNCtx c = gn_ctx(p);
    const int lt = LT >= 0 ? LT : p.loss_type;
    float col[3][8];   // A = sum dz, B = sum dz*xhat, X = sum xhat   (over this block's rows)
    float lsel = 0.f, lsq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { col[0][e] = 0.f; col[1][e] = 0.f; col[2][e] = 0.f; }
    float mean[8], rstd[8];
    gn_consts(p, c, mean, rstd);
    if (c.col_ok) {
        float gam[8], bet[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gam[e] = p.gamma[c.c0 + e]; bet[e] = p.beta[c.c0 + e]; }
        const T* y = reinterpret_cast<const T*>(p.y);
        const T* dout = reinterpret_cast<const T*>(p.dout);
        T* xo = reinterpret_cast<T*>(p.out);
        for (int t = c.t_lo + c.ty; t < c.t_hi; t += c.RL) {
            const long m = (long)c.b * p.T + t;
            float v[8], d[8];
            load8(y + m * p.ldy + c.c0, v);
            load8(dout + m * p.lddout + c.c0, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (v[e] - mean[e]) * rstd[e];
                const float z = xh * gam[e] + bet[e];
                float dz;
                if constexpr (FROM_LOSS) {
                    const float o = tanh_f(z);
                    const float df = o - d[e];
                    lsel += loss_val(lt, df);
                    lsq += df * df;
                    dz = loss_grad(lt, df) * (1.f - o * o);
                    v[e] = o;
                } else {
                    dz = d[e] * p.rscale * (ACT == 1 ? gelu_grad_f(z) : (ACT == 3 ? (z > 0.f ? 1.f : 0.f) : 1.f));
                }
                col[0][e] += dz;
                col[1][e] += dz * xh;
                col[2][e] += xh;
            }
            if constexpr (FROM_LOSS) {
                if (xo) store8(xo + m * p.ldout + c.c0, v);
            }
        }
    }
    if constexpr (FROM_LOSS) {
        // per-block loss partials (selected loss, squared error) -> lpart[block][2]; ew_recon_loss sums them in a fixed order
        __shared__ float sml[4];
        const float a = block_sum_fixed(lsel, sml);
        const float b2 = block_sum_fixed(lsq, sml);
        if (threadIdx.x == 0) { p.lpart[(long)block_linear() * 2] = a; p.lpart[(long)block_linear() * 2 + 1] = b2; }
    }
    if constexpr (TRAIN) gn_block_colsums<3>(p, c, col);      // per-block column sums; gn_bwd_finalize_kernel does the rest
}

// One block per (group, sample): combine the row-block partials of gn_bwd_reduce_kernel (A = sum dz, B = sum dz*xhat,
// X = sum xhat per column) in a fixed order:
//   sums2[b][g] = (s1, s2) = (sum_c gamma_c A_c, sum_c gamma_c B_c)                 (plain store: no zero-fill, no atomics)
//   ptot[b][0..2][c] = A_c, B_c, D_c = gscale * rstd * (gamma_c*A_c - T*s1/n - (s2/n)*X_c)   (D = column sum of dY, analytically)
// The sums over the samples (dbeta, dgamma, dbias) are taken later by ew_fin_affine.
__global__ __launch_bounds__(1024) void gn_bwd_finalize_kernel(const GNParams p, int RS) {
    __shared__ double smd[16];
    __shared__ float smr[3][1024];
    const int g = blockIdx.x, b = blockIdx.y;
    const float* part = p.part + ((long)b * RS * 3) * p.C;
    const int c_lo = g * p.Cg, c_hi = g == p.G - 1 ? p.C : (g + 1) * p.Cg;
    const int ncol = c_hi - c_lo;
    constexpr int MAXC = 12;                 // columns per thread kept in registers (recon head: 11876 / 1024 -> 12)
    float rA[MAXC], rB[MAXC], rX[MAXC];
    double s1 = 0.0, s2 = 0.0;
    const bool lanes = ncol <= 128;          // narrow groups (image conditioner): the row blocks are split over the threads too
    if (lanes) {
        int ncp = 1;
        while (ncp < ncol) ncp <<= 1;
        const int RLn = (int)blockDim.x / ncp, ci = threadIdx.x % ncp, rl = threadIdx.x / ncp;      // gn_finalize: 256 threads here
        float A = 0.f, Bv = 0.f, X = 0.f;
        if (ci < ncol)
            for (int r = rl; r < RS; r += RLn) {
                A += part[((long)r * 3 + 0) * p.C + c_lo + ci];
                Bv += part[((long)r * 3 + 1) * p.C + c_lo + ci];
                X += part[((long)r * 3 + 2) * p.C + c_lo + ci];
            }
        smr[0][threadIdx.x] = A; smr[1][threadIdx.x] = Bv; smr[2][threadIdx.x] = X;
        __syncthreads();
        A = Bv = X = 0.f;
        if (rl == 0 && ci < ncol) {
            for (int k = 0; k < RLn; ++k) { A += smr[0][k * ncp + ci]; Bv += smr[1][k * ncp + ci]; X += smr[2][k * ncp + ci]; }
            const float gm = p.gamma[c_lo + ci];
            s1 = (double)(gm * A); s2 = (double)(gm * Bv);
        }
        rA[0] = A; rB[0] = Bv; rX[0] = X;
    } else {
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {         // compile-time register indices (a run-time index would demote the arrays to scratch)
            const int c = c_lo + (int)threadIdx.x + k * 1024;
            float A = 0.f, Bv = 0.f, X = 0.f;
            if (c < c_hi) {
                for (int r = 0; r < RS; ++r) {
                    A += part[((long)r * 3 + 0) * p.C + c];
                    Bv += part[((long)r * 3 + 1) * p.C + c];
                    X += part[((long)r * 3 + 2) * p.C + c];
                }
                const float gm = p.gamma[c];
                s1 += (double)(gm * A); s2 += (double)(gm * Bv);
            }
            rA[k] = A; rB[k] = Bv; rX[k] = X;
        }
        for (int c = c_lo + (int)threadIdx.x + MAXC * 1024; c < c_hi; c += 1024) {      // wider groups than the register file holds
            float A = 0.f, Bv = 0.f;
            for (int r = 0; r < RS; ++r) {
                A += part[((long)r * 3 + 0) * p.C + c];
                Bv += part[((long)r * 3 + 1) * p.C + c];
            }
            const float gm = p.gamma[c];
            s1 += (double)(gm * A); s2 += (double)(gm * Bv);
        }
    }
    const double S1 = block_sum_f64(s1, smd), S2 = block_sum_f64(s2, smd);
    if (threadIdx.x == 0) {
        p.sums2[((long)b * p.G + g) * 2 + 0] = S1;
        p.sums2[((long)b * p.G + g) * 2 + 1] = S2;
    }
    const double n = (double)p.Cg * (double)p.T;
    const double sm_ = p.sums[((long)b * p.G + g) * 2 + 0], ss_ = p.sums[((long)b * p.G + g) * 2 + 1];
    const double mean = sm_ / n;
    double var = ss_ / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    const float m1 = (float)(S1 / n), m2 = (float)(S2 / n);
    float* pt = p.ptot + (long)b * 3 * p.C;
    if (lanes) {
        if (threadIdx.x < ncol) {
            const int c = c_lo + threadIdx.x;
            pt[c] = rA[0];
            pt[(long)p.C + c] = rB[0];
            pt[2L * p.C + c] = p.gscale * rstd * (p.gamma[c] * rA[0] - (float)p.T * m1 - m2 * rX[0]);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int c = c_lo + (int)threadIdx.x + k * 1024;
        if (c < c_hi) {
            pt[c] = rA[k];
            pt[(long)p.C + c] = rB[k];
            pt[2L * p.C + c] = p.gscale * rstd * (p.gamma[c] * rA[k] - (float)p.T * m1 - m2 * rX[k]);
        }
    }
    for (int c = c_lo + (int)threadIdx.x + MAXC * 1024; c < c_hi; c += 1024) {
        float A = 0.f, Bv = 0.f, X = 0.f;
        for (int r = 0; r < RS; ++r) {
            A += part[((long)r * 3 + 0) * p.C + c];
            Bv += part[((long)r * 3 + 1) * p.C + c];
            X += part[((long)r * 3 + 2) * p.C + c];
        }
        pt[c] = A;
        pt[(long)p.C + c] = Bv;
        pt[2L * p.C + c] = p.gscale * rstd * (p.gamma[c] * A - (float)p.T * m1 - m2 * X);
    }
}

// out[c] = sum over `rows` partial rows of part[row][c]; one block = 16 columns x 16 row-lanes
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* part, int rows, int C, float* out) {
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float a = 0.f;
    if (c < C)
        for (int r = rl; r < rows; r += 16) a += part[(long)r * C + c];
    __shared__ float sm[16][17];
    sm[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][cl];
        out[c] = t;
    }
}

// out[b*n + j] = scale * sum_{r < R} part[(b*R + r)*n + j]; one block = CL outputs x 256/CL row lanes (few outputs: many lanes),
// fixed summation order: lane-strided partial sums, then the lanes in index order
template <typename TI, typename TO, int CL>
__global__ __launch_bounds__(256) void rowsum_kernel(const TI* part, int R, int n, TO* out, double scale) {
    constexpr int RLN = 256 / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int j = blockIdx.x * CL + cl, b = blockIdx.y;
    double a = 0.0;
    if (j < n)
        for (int r = rl; r < R; r += RLN) a += (double)part[((long)b * R + r) * n + j];
    __shared__ double sm[RLN][CL + 1];
    sm[rl][cl] = a;
    __syncthreads();
    // two-level fixed-order combine: 16 threads per output sum RLN/16 lanes each, then one thread sums those
    __shared__ double sm2[16][CL + 1];
    if (rl < 16) {
        double t = 0.0;
        for (int k = rl; k < RLN; k += 16) t += sm[k][cl];
        sm2[rl][cl] = t;
    }
    __syncthreads();
    if (rl == 0 && j < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm2[k][cl];
        out[(long)b * n + j] = (TO)(t * scale);
    }
}
// one output, many rows (gradient-norm and KL partials): 1024 threads, four loads in flight per thread, fixed order
template <typename TI, typename TO>
__global__ __launch_bounds__(1024) void rowsum1_kernel(const TI* part, int R, TO* out, double scale) {
    __shared__ double sm16[16];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int r = threadIdx.x;
    for (; r + 3072 < R; r += 4096) { a0 += (double)part[r]; a1 += (double)part[r + 1024]; a2 += (double)part[r + 2048]; a3 += (double)part[r + 3072]; }
    for (; r < R; r += 1024) a0 += (double)part[r];
    const double t = block_sum_f64((a0 + a1) + (a2 + a3), sm16);
    if (threadIdx.x == 0) out[0] = (TO)(t * scale);
}
template <typename TI, typename TO>
static void rowsum_launch(const TI* part, int batches, int R, int n, TO* out, double scale, hipStream_t s) {
    if (n == 1 && batches == 1) hipLaunchKernelGGL((rowsum1_kernel<TI, TO>), dim3(1), dim3(1024), 0, s, part, R, out, scale);
    else if (n <= 1) hipLaunchKernelGGL((rowsum_kernel<TI, TO, 1>), dim3(n, batches), dim3(256), 0, s, part, R, n, out, scale);
    else if (n <= 4) hipLaunchKernelGGL((rowsum_kernel<TI, TO, 4>), dim3(cdiv_i(n, 4), batches), dim3(256), 0, s, part, R, n, out, scale);
    else hipLaunchKernelGGL((rowsum_kernel<TI, TO, 16>), dim3(cdiv_i(n, 16), batches), dim3(256), 0, s, part, R, n, out, scale);
}
int ew_rowsum(const float* part, int batches, int R, int n, float* out_f, double* out_d, double scale, hipStream_t s) {
    if (batches <= 0 || n <= 0) return 0;
    if (out_d) rowsum_launch<float, double>(part, batches, R, n, out_d, scale, s);
    else rowsum_launch<float, float>(part, batches, R, n, out_f, scale, s);
    return 0;
}
int ew_rowsum_d(const double* part, int R, int n, double* out_d, double scale, hipStream_t s) {
    if (n <= 0) return 0;
    rowsum_launch<double, double>(part, 1, R, n, out_d, scale, s);
    return 0;
}

// <G, W_eff> scalars: block i sums its item's per-block partials in a fixed order -> dst[0]
constexpr int FIN_MAX_ITEMS = 64;
struct FinDotArgs { int n; int pad; FinDot it[FIN_MAX_ITEMS]; };
__global__ __launch_bounds__(256) void fin_dot_kernel(const FinDotArgs a) {
    __shared__ float smw[4];
    const FinDot it = a.it[blockIdx.x];
    float v = 0.f;
    for (int i = threadIdx.x; i < it.count; i += 256) v += it.src[i];
    const float t = block_sum_fixed(v, smw);
    if (threadIdx.x == 0) it.dst[0] = t;
}
int ew_fin_dots(const FinDot* items, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += FIN_MAX_ITEMS) {
        FinDotArgs a; a.n = std::min(n - i0, FIN_MAX_ITEMS); a.pad = 0;
        for (int i = 0; i < a.n; ++i) a.it[i] = items[i0 + i];
        hipLaunchKernelGGL(fin_dot_kernel, dim3(a.n), dim3(256), 0, s, a);
    }
    return 0;
}
// GroupNorm affine + conv bias gradients: a block takes 32 columns of one item; row lane r (of 8) adds the per-sample / per-block
// totals r, r + 8, ... in order and the lanes are combined in lane order -- a fixed order.  (One thread per column walking all
// the rows left a 5120-column layer with 256 row blocks on 20 workgroups, 0.3 ms of pure load latency per step.)
struct FinAffineArgs { int n; int chunk0[FIN_MAX_ITEMS + 1]; FinAffine it[FIN_MAX_ITEMS]; };
constexpr int FIN_COLS = 32, FIN_LANES = 8;
__global__ __launch_bounds__(256) void fin_affine_kernel(const FinAffineArgs a) {
    __shared__ float sm[3][FIN_LANES][FIN_COLS];
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.chunk0[i + 1]) ++i;
    const FinAffine it = a.it[i];
    const int cl = threadIdx.x % FIN_COLS, rl = threadIdx.x / FIN_COLS;
    const int c = ((int)blockIdx.x - a.chunk0[i]) * FIN_COLS + cl;
    float A = 0.f, Bv = 0.f, D = 0.f;
    if (c < it.C) {
        if (it.arrays == 1) {             // [B][C] column-sum partials (bias gradient of a convolution without GroupNorm)
            for (int b = rl; b < it.B; b += FIN_LANES) A += it.ptot[(long)b * it.C + c];
        } else {
            for (int b = rl; b < it.B; b += FIN_LANES) {
                const float* pt = it.ptot + (long)b * 3 * it.C + c;
                A += pt[0]; Bv += pt[it.C]; D += pt[2L * it.C];
            }
        }
    }
    sm[0][rl][cl] = A; sm[1][rl][cl] = Bv; sm[2][rl][cl] = D;
    __syncthreads();
    if (rl != 0 || c >= it.C) return;
    A = Bv = D = 0.f;
#pragma unroll
    for (int r = 0; r < FIN_LANES; ++r) { A += sm[0][r][cl]; Bv += sm[1][r][cl]; D += sm[2][r][cl]; }
    if (it.dbeta) it.dbeta[c] = (it.accum ? it.dbeta[c] : 0.f) + A;
    if (it.dgamma) it.dgamma[c] = (it.accum ? it.dgamma[c] : 0.f) + Bv;
    if (it.dbias) it.dbias[c] = (it.accum ? it.dbias[c] : 0.f) + D;
}
int ew_fin_affine(const FinAffine* items, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += FIN_MAX_ITEMS) {
        FinAffineArgs a; a.n = std::min(n - i0, FIN_MAX_ITEMS);
        int ch = 0;
        for (int i = 0; i < a.n; ++i) { a.it[i] = items[i0 + i]; a.chunk0[i] = ch; ch += cdiv_i(a.it[i].C, FIN_COLS); }
        a.chunk0[a.n] = ch;
        if (ch > 0) hipLaunchKernelGGL(fin_affine_kernel, dim3(ch), dim3(256), 0, s, a);
    }
    return 0;
}

// dY = rstd * (gamma*dz - s1/n - xhat*s2/n) [* gscale]   (pure streaming pass: read y, dOut; write dY)
template <typename T, int ACT, bool FROM_LOSS, int LT = -1>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const GNParams p) {
    const GNCtx c = gn_ctx(p);
    const int lt = LT >= 0 ? LT : p.loss_type;
    float dotacc = 0.f;
    float mean[8], rstd[8], m1[8], m2[8];
    gn_consts<true>(p, c, mean, rstd, m1, m2);
    if (c.col_ok) {
        float gam[8], bet[8], cb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gam[e] = p.gamma[c.c0 + e];
            bet[e] = p.beta[c.c0 + e];
            cb[e] = p.cbias ? p.cbias[c.c0 + e] : 0.f;
        }
        const T* y = reinterpret_cast<const T*>(p.y);
        const T* dout = reinterpret_cast<const T*>(p.dout);
        T* dy = reinterpret_cast<T*>(p.out);
        for (int t = c.t_lo + c.ty; t < c.t_hi; t += 2 * c.RL) {      // two rows per round, all four loads first
            const bool two = t + c.RL < c.t_hi;
            const long r0 = (long)c.b * p.T + t, r1 = two ? r0 + c.RL : r0;
            Raw8<T> ry0, ry1, rd0, rd1;
            raw_load(y + r0 * p.ldy + c.c0, ry0);
            raw_load(dout + r0 * p.lddout + c.c0, rd0);
            raw_load(y + r1 * p.ldy + c.c0, ry1);
            raw_load(dout + r1 * p.lddout + c.c0, rd1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                float v[8], d[8];
                raw_unpack(u ? ry1 : ry0, v);
                raw_unpack(u ? rd1 : rd0, d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (v[e] - mean[e]) * rstd[e];
                    const float z = xh * gam[e] + bet[e];
                    float dz;
                    if constexpr (FROM_LOSS) {
                        const float o = tanh_f(z);
                        dz = loss_grad(lt, o - d[e]) * (1.f - o * o);
                    } else {
                        dz = d[e] * p.rscale * (ACT == 1 ? gelu_grad_f(z) : (ACT == 3 ? (z > 0.f ? 1.f : 0.f) : 1.f));
                    }
                    const float r = rstd[e] * (gam[e] * dz - m1[e] - xh * m2[e]) * p.gscale;
                    dotacc += r * (v[e] - cb[e]);          // dY * (conv output without bias) -> <G, W_eff>
                    v[e] = r;
                }
                store8(dy + (u ? r1 : r0) * p.ldout + c.c0, v);
            }
        }
    }
    if (p.cdot_part) block_store_partial(dotacc, p.cdot_part);
}

// ------------------------------------------------------------------------------------------
// Small layers: ONE block per (group, sample) owns the whole T x Cg slab and keeps it in registers (every thread at
// most GN_FUSED_ITERS rows of one 8-wide column vector), so statistics + normalise (forward) and reduce + finalize +
// dY (backward) are one launch and one pass over memory each instead of two and three launches: these layers move a
// few MB and were bound by launch latency and dependent-load latency, not by bandwidth.  All loads of a thread are
// issued before the first use.  Thread layout: tx = column vector inside the group (CVg = next power of two >= Cg/8),
// ty = row lane.  Results match the multi-kernel path up to summation order (float partials per thread, fp64 across).
// ------------------------------------------------------------------------------------------
constexpr int GN_FUSED_ITERS = 16;
struct GNSlab {
    int g, b, nv, CVg, RL, tx, ty, c0;
    bool col_ok;
};
__device__ __forceinline__ GNSlab gn_slab(const GNParams& p) {
    GNSlab c;
    c.g = blockIdx.x; c.b = blockIdx.y;
    c.nv = p.Cg >> 3;
    c.CVg = p.CV;                       // host: next power of two >= Cg / 8 (<= 256)
    c.RL = 256 / c.CVg;
    c.tx = threadIdx.x % c.CVg; c.ty = threadIdx.x / c.CVg;
    c.col_ok = c.tx < c.nv;
    c.c0 = c.g * p.Cg + c.tx * 8;
    return c;
}
// out = [res + rscale *] act(gn(y)), statistics included; p.sums[(b*G+g)*2 + {0,1}] = sum, sum of squares (stored)
template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_fwd_fused_kernel(const GNParams p) {
    __shared__ double sm4[16];
    const GNSlab c = gn_slab(p);
    const T* y = reinterpret_cast<const T*>(p.y) + (long)c.b * p.T * p.ldy + c.c0;
    const T* res = p.res ? reinterpret_cast<const T*>(p.res) + (long)c.b * p.T * p.ldres + c.c0 : nullptr;
    Raw8<T> ry[GN_FUSED_ITERS], rr[GN_FUSED_ITERS];
#pragma unroll
    for (int i = 0; i < GN_FUSED_ITERS; ++i) {
        const int t = c.ty + i * c.RL;
        if (c.col_ok && t < p.T) {
            raw_load(y + (long)t * p.ldy, ry[i]);
            if (res) raw_load(res + (long)t * p.ldres, rr[i]);
        }
    }
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < GN_FUSED_ITERS; ++i) {
        const int t = c.ty + i * c.RL;
        if (c.col_ok && t < p.T) {
            float v[8];
            raw_unpack(ry[i], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a += v[e]; q += v[e] * v[e]; }
        }
    }
    const double S = block_sum_f64((double)a, sm4);
    const double SS = block_sum_f64((double)q, sm4);
    if (threadIdx.x == 0) {
        p.sums[((long)c.b * p.G + c.g) * 2 + 0] = S;
        p.sums[((long)c.b * p.G + c.g) * 2 + 1] = SS;
    }
    const double n = (double)p.Cg * (double)p.T;
    const double md = S / n;
    double var = SS / n - md * md;
    if (var < 0.0) var = 0.0;
    const float mean = (float)md, rstd = (float)(1.0 / sqrt(var + 1e-5));
    if (!c.col_ok) return;
    float ka[8], kb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float g = p.gamma[c.c0 + e];
        ka[e] = rstd * g;
        kb[e] = p.beta[c.c0 + e] - mean * rstd * g;
    }
    T* out = reinterpret_cast<T*>(p.out) + (long)c.b * p.T * p.ldout + c.c0;
#pragma unroll
    for (int i = 0; i < GN_FUSED_ITERS; ++i) {
        const int t = c.ty + i * c.RL;
        if (t < p.T) {
            float v[8], r[8];
            raw_unpack(ry[i], v);
            if (res) raw_unpack(rr[i], r);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = act_apply(ACT, v[e] * ka[e] + kb[e]);
                v[e] = res ? r[e] + p.rscale * f : f;
            }
            store8(out + (long)t * p.ldout, v);
        }
    }
}

// gn_bwd_reduce + gn_bwd_finalize + gn_bwd_apply of one (group, sample) slab (stored incoming gradient, ACT in {0,1,3}).
// 512 threads (the pass is VALU-heavy: erf/exp of the GELU derivative, computed once and kept in registers as dz);
// column sums: butterfly over the lanes of a wave that share a column vector, then across the 8 waves through LDS.
constexpr int GN_BWD_THREADS = 512, GN_BWD_ITERS = 8, GN_BWD_MAX_CV = 32;
template <typename T, int ACT>
__global__ __launch_bounds__(GN_BWD_THREADS) void gn_bwd_fused_kernel(const GNParams p) {
    constexpr int NW = GN_BWD_THREADS / 64;
    __shared__ float smc[3][NW][GN_BWD_MAX_CV * 8];
    __shared__ float smw[2 * NW];
    __shared__ float smk[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int nv = p.Cg >> 3, CVg = p.CV, RL = GN_BWD_THREADS / CVg;
    const int tx = threadIdx.x % CVg, ty = threadIdx.x / CVg, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool col_ok = tx < nv;
    const int c0 = g * p.Cg + tx * 8;
    const T* y = reinterpret_cast<const T*>(p.y) + (long)b * p.T * p.ldy + c0;
    const T* dout = reinterpret_cast<const T*>(p.dout) + (long)b * p.T * p.lddout + c0;
    Raw8<T> ry[GN_BWD_ITERS], rd[GN_BWD_ITERS];
#pragma unroll
    for (int i = 0; i < GN_BWD_ITERS; ++i) {
        const int t = ty + i * RL;
        if (col_ok && t < p.T) {
            raw_load(y + (long)t * p.ldy, ry[i]);
            raw_load(dout + (long)t * p.lddout, rd[i]);
        }
    }
    const double n = (double)p.Cg * (double)p.T;
    if (threadIdx.x == 0) {
        const double s = p.sums[((long)b * p.G + g) * 2 + 0], ss = p.sums[((long)b * p.G + g) * 2 + 1];
        const double m = s / n;
        double var = ss / n - m * m;
        if (var < 0.0) var = 0.0;
        smk[0] = (float)m;
        smk[1] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const float mean = smk[0], rstd = smk[1];
    float gam[8], bet[8];
    float col[3][8];
    float dz[GN_BWD_ITERS][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { col[0][e] = 0.f; col[1][e] = 0.f; col[2][e] = 0.f; gam[e] = 0.f; bet[e] = 0.f; }
    if (col_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { gam[e] = p.gamma[c0 + e]; bet[e] = p.beta[c0 + e]; }
    }
#pragma unroll
    for (int i = 0; i < GN_BWD_ITERS; ++i) {
        const int t = ty + i * RL;
        if (col_ok && t < p.T) {
            float v[8], d[8];
            raw_unpack(ry[i], v);
            raw_unpack(rd[i], d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (v[e] - mean) * rstd;
                const float z = xh * gam[e] + bet[e];
                const float q = d[e] * p.rscale * (ACT == 1 ? gelu_grad_f(z) : (ACT == 3 ? (z > 0.f ? 1.f : 0.f) : 1.f));
                dz[i][e] = q;
                col[0][e] += q;
                col[1][e] += q * xh;
                col[2][e] += xh;
            }
        }
    }
    // column sums: lanes l, l + CVg, l + 2 CVg ... of a wave hold the same column vector
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = col[k][e];
            for (int o = 32; o >= CVg; o >>= 1) v += __shfl_xor(v, o, 64);
            col[k][e] = v;
        }
    if (lane < CVg) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) smc[k][wave][lane * 8 + e] = col[k][e];
    }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    const bool owner = threadIdx.x < CVg && col_ok;          // wave 0, one lane per column vector (tx == lane)
    if (owner) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = 0.f;
                for (int w = 0; w < NW; ++w) v += smc[k][w][lane * 8 + e];
                col[k][e] = v;
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += gam[e] * col[0][e]; s2 += gam[e] * col[1][e]; }
    }
    if (wave == 0) {
        const float w1 = wave_sum(s1), w2 = wave_sum(s2);
        if (lane == 0) { smw[0] = w1; smw[1] = w2; }
    }
    __syncthreads();
    s1 = smw[0];
    s2 = smw[1];
    if (threadIdx.x == 0) {
        p.sums2[((long)b * p.G + g) * 2 + 0] = (double)s1;
        p.sums2[((long)b * p.G + g) * 2 + 1] = (double)s2;
    }
    const float m1 = (float)((double)s1 / n), m2 = (float)((double)s2 / n);
    if (owner) {      // per-sample column totals (A, B, D): summed over the samples by ew_fin_affine
        float* pt = p.ptot + (long)b * 3 * p.C + c0;
        float dcol[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dcol[e] = p.gscale * rstd * (gam[e] * col[0][e] - (float)p.T * m1 - m2 * col[2][e]);
        store8(pt, col[0]);
        store8(pt + p.C, col[1]);
        store8(pt + 2L * p.C, dcol);
    }
    // dY = rstd * (gamma*dz - s1/n - xhat*s2/n) * gscale, and <G, W_eff> += sum dY * (y - conv bias)
    float dotacc = 0.f;
    if (col_ok) {
        float cb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cb[e] = p.cbias ? p.cbias[c0 + e] : 0.f;
        T* dy = reinterpret_cast<T*>(p.out) + (long)b * p.T * p.ldout + c0;
#pragma unroll
        for (int i = 0; i < GN_BWD_ITERS; ++i) {
            const int t = ty + i * RL;
            if (t < p.T) {
                float v[8];
                raw_unpack(ry[i], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (v[e] - mean) * rstd;
                    const float r = rstd * (gam[e] * dz[i][e] - m1 - xh * m2) * p.gscale;
                    dotacc += r * (v[e] - cb[e]);
                    v[e] = r;
                }
                store8(dy + (long)t * p.ldout, v);
            }
        }
    }
    if (p.cdot_part) {
        const float w = wave_sum(dotacc);
        __syncthreads();
        if (lane == 0) smw[wave] = w;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
            for (int k = 0; k < NW; ++k) tot += smw[k];
            p.cdot_part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
        }
    }
}

// activation without GroupNorm: MODE 0: out = gelu(y); MODE 1: out = dout*rscale*gelu'(y) (+ colsum -> dbias)
//                               MODE 2: column sums of y only (-> dbias)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void act_kernel(const GNParams p) {
    const GNCtx c = gn_ctx(p);
    float colD[8];
    float dotacc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) colD[e] = 0.f;
    if (c.col_ok) {
        const T* y = reinterpret_cast<const T*>(p.y);
        const T* dout = reinterpret_cast<const T*>(p.dout);
        T* out = reinterpret_cast<T*>(p.out);
        float cb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cb[e] = (MODE != 0 && p.cbias) ? p.cbias[c.c0 + e] : 0.f;
        for (int t = c.t_lo + c.ty; t < c.t_hi; t += c.RL) {
            const long m = (long)c.b * p.T + t;
            float v[8], d[8];
            load8(y + m * p.ldy + c.c0, v);
            if constexpr (MODE == 1) load8(dout + m * p.lddout + c.c0, d);
            if constexpr (MODE == 2) { if (p.yf32) load8(p.yf32 + m * p.ldyf + c.c0, d); }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (MODE == 0) v[e] = gelu_f(v[e]);
                else if constexpr (MODE == 1) {
                    const float yv = v[e];
                    v[e] = d[e] * p.rscale * gelu_grad_f(yv);
                    dotacc += v[e] * (yv - cb[e]);
                } else {
                    if (p.yf32) dotacc += v[e] * (d[e] - cb[e]);     // v = dY, d = conv output (fp32)
                }
                colD[e] += v[e];
            }
            if constexpr (MODE != 2) store8(out + m * p.ldout + c.c0, v);
        }
    }
    if (MODE != 0 && p.cdot_part) block_store_partial(dotacc, p.cdot_part);
    if (MODE != 0 && p.part) {
        float col[1][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) col[0][e] = colD[e];
        gn_block_colsums<1>(p, c, col);
    }
}

#define GN_LAUNCH(KERN, P, S, ...)                                         \
    do {                                                                   \
        GNGeom g_ = gn_geom((P).B, (P).T, (P).C);                          \
        (P).CV = g_.CV;                                                    \
        hipLaunchKernelGGL(KERN, g_.grid, dim3(256), 0, S, P, ##__VA_ARGS__); \
    } while (0)
// reduce-type kernels write per-block column sums: coarser row split keeps that workspace traffic small
constexpr int GN_REDUCE_TARGET = 768;
#define GN_LAUNCH_R(KERN, P, S)                                            \
    do {                                                                   \
        GNGeom g_ = gn_geom((P).B, (P).T, (P).C, GN_REDUCE_TARGET);        \
        (P).CV = g_.CV;                                                    \
        hipLaunchKernelGGL(KERN, g_.grid, dim3(256), 0, S, P);             \
    } while (0)

static int grid_total(const dim3& g) { return (int)(g.x * g.y * g.z); }
// Workspace layout inside p.part (floats): [per-block column sums B*RS*3*C | per-sample totals B*3*C | <G,W_eff> block partials |
// loss block partials x2]; the forward statistics pass reuses the front for its per-block group sums.
struct GNWork { size_t ptot, dots, lpart, total; int nblk; };
static GNWork gn_work(int B, int T, int C) {
    const GNGeom gr = gn_geom(B, T, C, GN_REDUCE_TARGET), ga = gn_geom(B, T, C);
    GNWork w;
    size_t front = (size_t)B * gr.rowsplit * 3 * C;
    front = std::max(front, (size_t)grid_total(gr.grid) * SGV_GN_MAX_GROUPS * 2);
    w.ptot = (front + 3) & ~(size_t)3;
    w.dots = w.ptot + (size_t)B * 3 * C;
    w.nblk = std::max(std::max(grid_total(gr.grid), grid_total(ga.grid)), SGV_GN_MAX_GROUPS * B);
    w.lpart = w.dots + w.nblk;
    w.total = w.lpart + 2 * (size_t)w.nblk;
    return w;
}
size_t ew_gn_part_floats(int B, int T, int C) { return gn_work(B, T, C).total; }
int ew_act_part_rows(int B, int T, int C) { return B * gn_geom(B, T, C, GN_REDUCE_TARGET).rowsplit; }
int ew_gn_max_blocks(int B, int T, int C) { return gn_work(B, T, C).nblk; }
// deferred mode (p.ptot / p.cdot_part given): only report the partial count; else sum the partials here
static void gn_fin_immediate(const GNParams& p, bool own_ptot, bool own_dots, int nblk, hipStream_t s) {
    if (p.cdot_blocks) *p.cdot_blocks = nblk;
    if (own_ptot) { FinAffine it = {p.ptot, p.dbeta, p.dgamma, p.dbias, p.C, p.B, p.accum_affine, 0}; ew_fin_affine(&it, 1, s); }
    if (own_dots) { FinDot d = {p.cdot_part, p.cdot, nblk, 0}; ew_fin_dots(&d, 1, s); }
}
static void gn_finalize(GNParams p, hipStream_t s) {
    GNGeom g_ = gn_geom(p.B, p.T, p.C, GN_REDUCE_TARGET);
    // narrow groups (the kernel's `lanes` path, <= 128 columns): a few dozen row blocks x a few columns per (group, sample) --
    // 256 threads cover them and the block's two reductions cost a quarter of the barriers' wave count
    const int ncol_max = p.C - (p.G - 1) * p.Cg > p.Cg ? p.C - (p.G - 1) * p.Cg : p.Cg;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(p.G, p.B), dim3(ncol_max <= 128 ? 256 : 1024), 0, s, p, g_.rowsplit);
}
// p.sums[b][g] = (sum, sum of squares), overwritten; p.part = workspace (ew_gn_part_floats)
int ew_gn_stats(int dtype, GNParams p, hipStream_t s) {
    if (!p.part) return -1;
    if (dtype == 1) GN_LAUNCH_R((gn_stats_kernel<bf16_t>), p, s);
    else GN_LAUNCH_R((gn_stats_kernel<float>), p, s);
    const GNGeom g_ = gn_geom(p.B, p.T, p.C, GN_REDUCE_TARGET);
    return ew_rowsum(p.part, p.B, (int)(g_.grid.x * g_.grid.y), p.G * 2, nullptr, p.sums, 1.0, s);
}
int ew_gn_apply(int dtype, int act, GNParams p, hipStream_t s) {
    if (dtype == 1) {
        if (act == 1) GN_LAUNCH((gn_apply_kernel<bf16_t, 1>), p, s);
        else if (act == 2) GN_LAUNCH((gn_apply_kernel<bf16_t, 2>), p, s);
        else if (act == 3) GN_LAUNCH((gn_apply_kernel<bf16_t, 3>), p, s);
        else GN_LAUNCH((gn_apply_kernel<bf16_t, 0>), p, s);
    } else {
        if (act == 1) GN_LAUNCH((gn_apply_kernel<float, 1>), p, s);
        else if (act == 2) GN_LAUNCH((gn_apply_kernel<float, 2>), p, s);
        else if (act == 3) GN_LAUNCH((gn_apply_kernel<float, 3>), p, s);
        else GN_LAUNCH((gn_apply_kernel<float, 0>), p, s);
    }
    return 0;
}
int ew_gn_tail(int dtype, GNParams p, GNTail t, hipStream_t s) {
    if (!p.y || !p.out || !t.y2 || !p.sums || (t.cscale ? 0 : (!t.sums2 || !t.gamma2 || !t.beta2))) return -1;
    if (dtype == 1) {
        if (t.cscale) GN_LAUNCH((gn_tail_kernel<bf16_t, true>), p, s, t);
        else GN_LAUNCH((gn_tail_kernel<bf16_t, false>), p, s, t);
    } else {
        if (t.cscale) GN_LAUNCH((gn_tail_kernel<float, true>), p, s, t);
        else GN_LAUNCH((gn_tail_kernel<float, false>), p, s, t);
    }
    return 0;
}
// GroupNorm backward for act in {0 none, 1 gelu, 3 relu}: reduce + finalize, then the streaming dY pass
int ew_gn_bwd_reduce_act(int dtype, int act, GNParams p, hipStream_t s) {
    if (dtype == 1) {
        if (act == 1) GN_LAUNCH_R((gn_bwd_reduce_kernel<bf16_t, 1, false, true>), p, s);
        else if (act == 3) GN_LAUNCH_R((gn_bwd_reduce_kernel<bf16_t, 3, false, true>), p, s);
        else GN_LAUNCH_R((gn_bwd_reduce_kernel<bf16_t, 0, false, true>), p, s);
    } else {
        if (act == 1) GN_LAUNCH_R((gn_bwd_reduce_kernel<float, 1, false, true>), p, s);
        else if (act == 3) GN_LAUNCH_R((gn_bwd_reduce_kernel<float, 3, false, true>), p, s);
        else GN_LAUNCH_R((gn_bwd_reduce_kernel<float, 0, false, true>), p, s);
    }
    gn_finalize(p, s);
    return 0;
}
int ew_gn_bwd_apply_act(int dtype, int act, GNParams p, hipStream_t s) {
    if (dtype == 1) {
        if (act == 1) GN_LAUNCH((gn_bwd_apply_kernel<bf16_t, 1, false>), p, s);
        else if (act == 3) GN_LAUNCH((gn_bwd_apply_kernel<bf16_t, 3, false>), p, s);
        else GN_LAUNCH((gn_bwd_apply_kernel<bf16_t, 0, false>), p, s);
    } else {
        if (act == 1) GN_LAUNCH((gn_bwd_apply_kernel<float, 1, false>), p, s);
        else if (act == 3) GN_LAUNCH((gn_bwd_apply_kernel<float, 3, false>), p, s);
        else GN_LAUNCH((gn_bwd_apply_kernel<float, 0, false>), p, s);
    }
    return 0;
}
// one launch per direction for small slabs (SGV_GN_FUSED=0 restores the multi-kernel path for A/B runs)
static bool gn_fused_ok(const GNParams& p) {
    static const int on = getenv("SGV_GN_FUSED") ? atoi(getenv("SGV_GN_FUSED")) : 1;
    if (!on || p.Cg % 8 || p.Cg / 8 > 256 || p.G > SGV_GN_MAX_GROUPS) return false;
    int cv = 1;
    while (cv < p.Cg / 8) cv <<= 1;
    return (p.T + 256 / cv - 1) / (256 / cv) <= GN_FUSED_ITERS;      // rows per thread
}
static int gn_fused_cv(const GNParams& p) {
    int cv = 1;
    while (cv < p.Cg / 8) cv <<= 1;
    return cv;
}
static bool gn_fused_bwd_ok(const GNParams& p) {
    static const int on = getenv("SGV_GN_FUSED") ? atoi(getenv("SGV_GN_FUSED")) : 1;
    if (!on || p.Cg % 8 || p.G > SGV_GN_MAX_GROUPS) return false;
    const int cv = gn_fused_cv(p);
    return cv <= GN_BWD_MAX_CV && (p.T + GN_BWD_THREADS / cv - 1) / (GN_BWD_THREADS / cv) <= GN_BWD_ITERS;
}
#define GN_FUSED_LAUNCH_N(KERN, NT, P, S)                                                    \
    do {                                                                                     \
        (P).CV = gn_fused_cv(P);                                                             \
        hipLaunchKernelGGL(KERN, dim3((P).G, (P).B), dim3(NT), 0, S, P);                     \
    } while (0)
#define GN_FUSED_LAUNCH(KERN, P, S) GN_FUSED_LAUNCH_N(KERN, 256, P, S)
// statistics + normalise (p.sums is overwritten); the multi-kernel path needs the workspace p.part
int ew_gn_fwd(int dtype, int act, GNParams p, hipStream_t s) {
    if (!gn_fused_ok(p) || !p.out) {
        GNParams q = p;
        if (ew_gn_stats(dtype, q, s)) return -1;
        return p.out ? ew_gn_apply(dtype, act, p, s) : 0;
    }
    if (dtype == 1) {
        if (act == 1) GN_FUSED_LAUNCH((gn_fwd_fused_kernel<bf16_t, 1>), p, s);
        else if (act == 2) GN_FUSED_LAUNCH((gn_fwd_fused_kernel<bf16_t, 2>), p, s);
        else if (act == 3) GN_FUSED_LAUNCH((gn_fwd_fused_kernel<bf16_t, 3>), p, s);
        else GN_FUSED_LAUNCH((gn_fwd_fused_kernel<bf16_t, 0>), p, s);
    } else {
        if (act == 1) GN_FUSED_LAUNCH((gn_fwd_fused_kernel<float, 1>), p, s);
        else if (act == 2) GN_FUSED_LAUNCH((gn_fwd_fused_kernel<float, 2>), p, s);
        else if (act == 3) GN_FUSED_LAUNCH((gn_fwd_fused_kernel<float, 3>), p, s);
        else GN_FUSED_LAUNCH((gn_fwd_fused_kernel<float, 0>), p, s);
    }
    return 0;
}
// reduce + finalize + dY for act in {0 none, 1 gelu, 3 relu}; p carries both the reduce and the apply arguments.
// p.part = workspace (ew_gn_part_floats).  p.ptot / p.cdot_part given: deferred mode (the caller runs ew_fin_affine /
// ew_fin_dots later); else dbeta / dgamma / dbias / cdot[0] are final when this returns.
int ew_gn_bwd(int dtype, int act, GNParams p, hipStream_t s) {
    if (!p.part) return -1;
    const GNWork w = gn_work(p.B, p.T, p.C);
    const bool own_ptot = !p.ptot, own_dots = p.cdot && !p.cdot_part;
    if (own_ptot) p.ptot = p.part + w.ptot;
    if (own_dots) p.cdot_part = p.part + w.dots;
    if (!p.cdot) p.cdot_part = nullptr;
    if (!gn_fused_bwd_ok(p)) {
        GNParams q = p;
        q.out = nullptr; q.cdot = nullptr; q.cdot_part = nullptr; q.cbias = nullptr;
        ew_gn_bwd_reduce_act(dtype, act, q, s);
        ew_gn_bwd_apply_act(dtype, act, p, s);
        gn_fin_immediate(p, own_ptot, own_dots, grid_total(gn_geom(p.B, p.T, p.C).grid), s);
        return 0;
    }
    if (dtype == 1) {
        if (act == 1) GN_FUSED_LAUNCH_N((gn_bwd_fused_kernel<bf16_t, 1>), GN_BWD_THREADS, p, s);
        else if (act == 3) GN_FUSED_LAUNCH_N((gn_bwd_fused_kernel<bf16_t, 3>), GN_BWD_THREADS, p, s);
        else GN_FUSED_LAUNCH_N((gn_bwd_fused_kernel<bf16_t, 0>), GN_BWD_THREADS, p, s);
    } else {
        if (act == 1) GN_FUSED_LAUNCH_N((gn_bwd_fused_kernel<float, 1>), GN_BWD_THREADS, p, s);
        else if (act == 3) GN_FUSED_LAUNCH_N((gn_bwd_fused_kernel<float, 3>), GN_BWD_THREADS, p, s);
        else GN_FUSED_LAUNCH_N((gn_bwd_fused_kernel<float, 0>), GN_BWD_THREADS, p, s);
    }
    gn_fin_immediate(p, own_ptot, own_dots, p.G * p.B, s);
    return 0;
}
// tanh + loss (+ backward reduction): p.loss_sums[0..1] = selected loss sum, squared-error sum (overwritten);
// train: p.sums2 and the unit-weight dbeta / dgamma / dbias are final on return.  p.part = workspace.
int ew_recon_loss(int dtype, int train, GNParams p, hipStream_t s) {
    if (!p.part) return -1;
    const GNWork w = gn_work(p.B, p.T, p.C);
    p.lpart = p.part + w.lpart;
    p.ptot = p.part + w.ptot;
    if (dtype == 1 && train) {
        if (p.loss_type == 0) GN_LAUNCH_R((gn_bwd_reduce_kernel<bf16_t, 2, true, true, 0>), p, s);
        else if (p.loss_type == 1) GN_LAUNCH_R((gn_bwd_reduce_kernel<bf16_t, 2, true, true, 1>), p, s);
        else GN_LAUNCH_R((gn_bwd_reduce_kernel<bf16_t, 2, true, true, 2>), p, s);
    } else if (dtype == 1) {
        if (p.loss_type == 0) GN_LAUNCH((gn_bwd_reduce_kernel<bf16_t, 2, true, false, 0>), p, s);
        else if (p.loss_type == 1) GN_LAUNCH((gn_bwd_reduce_kernel<bf16_t, 2, true, false, 1>), p, s);
        else GN_LAUNCH((gn_bwd_reduce_kernel<bf16_t, 2, true, false, 2>), p, s);
    } else {
        if (train) GN_LAUNCH_R((gn_bwd_reduce_kernel<float, 2, true, true>), p, s);
        else GN_LAUNCH((gn_bwd_reduce_kernel<float, 2, true, false>), p, s);
    }
    const int nblk = grid_total(train ? gn_geom(p.B, p.T, p.C, GN_REDUCE_TARGET).grid : gn_geom(p.B, p.T, p.C).grid);
    ew_rowsum(p.lpart, 1, nblk, 2, nullptr, p.loss_sums, 1.0, s);
    if (train) {
        gn_finalize(p, s);
        FinAffine it = {p.ptot, p.dbeta, p.dgamma, p.dbias, p.C, p.B, 0, 0};
        ew_fin_affine(&it, 1, s);
    }
    return 0;
}
int ew_recon_bwd_apply(int dtype, GNParams p, hipStream_t s) {
    const bool own_dots = p.cdot && !p.cdot_part;
    if (own_dots) { if (!p.part) return -1; p.cdot_part = p.part + gn_work(p.B, p.T, p.C).dots; }
    if (!p.cdot) p.cdot_part = nullptr;
    if (dtype == 1) {
        if (p.loss_type == 0) GN_LAUNCH((gn_bwd_apply_kernel<bf16_t, 2, true, 0>), p, s);
        else if (p.loss_type == 1) GN_LAUNCH((gn_bwd_apply_kernel<bf16_t, 2, true, 1>), p, s);
        else GN_LAUNCH((gn_bwd_apply_kernel<bf16_t, 2, true, 2>), p, s);
    } else GN_LAUNCH((gn_bwd_apply_kernel<float, 2, true>), p, s);
    gn_fin_immediate(p, false, own_dots, grid_total(gn_geom(p.B, p.T, p.C).grid), s);
    return 0;
}
int ew_act(int dtype, int mode, GNParams p, hipStream_t s) {
    float* const ws = p.part;
    const bool own_dots = mode != 0 && p.cdot && !p.cdot_part;
    if (own_dots) { if (!ws) return -1; p.cdot_part = ws + gn_work(p.B, p.T, p.C).dots; }
    if (mode == 0 || !p.cdot) p.cdot_part = nullptr;
    if (mode == 0 || !p.dbias) p.part = nullptr;
    if (p.part) {
        if (dtype == 1) { if (mode == 1) GN_LAUNCH_R((act_kernel<bf16_t, 1>), p, s); else GN_LAUNCH_R((act_kernel<bf16_t, 2>), p, s); }
        else { if (mode == 1) GN_LAUNCH_R((act_kernel<float, 1>), p, s); else GN_LAUNCH_R((act_kernel<float, 2>), p, s); }
        // combine the per-block column sums into the bias gradient
        GNGeom g_ = gn_geom(p.B, p.T, p.C, GN_REDUCE_TARGET);
        if (!p.defer_colsum) hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv_i(p.C, 16)), dim3(256), 0, s, p.part, p.B * g_.rowsplit, p.C, p.dbias);
        gn_fin_immediate(p, false, own_dots, grid_total(g_.grid), s);
        return 0;
    }
    if (dtype == 1) {
        if (mode == 0) GN_LAUNCH((act_kernel<bf16_t, 0>), p, s);
        else if (mode == 1) GN_LAUNCH((act_kernel<bf16_t, 1>), p, s);
        else GN_LAUNCH((act_kernel<bf16_t, 2>), p, s);
    } else {
        if (mode == 0) GN_LAUNCH((act_kernel<float, 0>), p, s);
        else if (mode == 1) GN_LAUNCH((act_kernel<float, 1>), p, s);
        else GN_LAUNCH((act_kernel<float, 2>), p, s);
    }
    if (mode != 0) gn_fin_immediate(p, false, own_dots, grid_total(gn_geom(p.B, p.T, p.C).grid), s);
    return 0;
}

// ------------------------------------------------------------------------------------------
// strided elementwise helpers on [rows][C] maps (C % 8 == 0)
// ------------------------------------------------------------------------------------------
// out = a (+ b) (+ c): any of b, c may be null; all same dtype T
template <typename T>
__global__ __launch_bounds__(256) void add3_kernel(const T* a, long lda, const T* b, long ldb, const T* c, long ldc,
                                                  T* out, long ldo, int rows, int C) {
    const int nv = C / 8;
    const long total = (long)rows * nv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / nv), cv = (int)(i - (long)r * nv);
        float v[8], w[8];
        load8(a + (long)r * lda + cv * 8, v);
        if (b) { load8(b + (long)r * ldb + cv * 8, w);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += w[e]; }
        if (c) { load8(c + (long)r * ldc + cv * 8, w);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += w[e]; }
        store8(out + (long)r * ldo + cv * 8, v);
    }
}
int ew_add3(int dtype, const void* a, long lda, const void* b, long ldb, const void* c, long ldc, void* out, long ldo,
            int rows, int C, hipStream_t s) {
    long total = (long)rows * (C / 8);
    int blocks = cdiv_i(total, 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (dtype == 1)
        hipLaunchKernelGGL((add3_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)a, lda, (const bf16_t*)b, ldb,
                           (const bf16_t*)c, ldc, (bf16_t*)out, ldo, rows, C);
    else
        hipLaunchKernelGGL((add3_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)a, lda, (const float*)b, ldb,
                           (const float*)c, ldc, (float*)out, ldo, rows, C);
    return 0;
}

// batched transpose with cast: dst[b][j][i] = src[b][i][j]   (src rows I, cols J contiguous)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* src, TO* dst, int I, int J, long lds_, long ldd,
                                                        long sbatch, long dbatch) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        tile[r][tx] = (i < I && j < J) ? to_f32(src[(long)b * sbatch + (long)i * lds_ + j]) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        if (i < I && j < J) dst[(long)b * dbatch + (long)j * ldd + i] = from_f32<TO>(tile[tx][r]);
    }
}
int ew_transpose(int src_dtype, int dst_dtype, const void* src, void* dst, int Bn, int I, int J, long lds_, long ldd,
                 long sbatch, long dbatch, hipStream_t s) {
    dim3 grid(cdiv_i(J, 32), cdiv_i(I, 32), Bn);
    if (src_dtype == 0 && dst_dtype == 0)
        hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, s, (const float*)src, (float*)dst, I, J, lds_, ldd, sbatch, dbatch);
    else if (src_dtype == 0 && dst_dtype == 1)
        hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, dim3(256), 0, s, (const float*)src, (bf16_t*)dst, I, J, lds_, ldd, sbatch, dbatch);
    else if (src_dtype == 1 && dst_dtype == 0)
        hipLaunchKernelGGL((transpose_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t*)src, (float*)dst, I, J, lds_, ldd, sbatch, dbatch);
    else
        hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, I, J, lds_, ldd, sbatch, dbatch);
    return 0;
}

// standard normals into an fp32 buffer [rows][per_row] (Philox keyed by seed/stream).  Row b is sample b of this rank's batch =
// sample b * world + rank of the GLOBAL batch (shards are r::world), and its elements take the counters
// (global row) * per_row + j: a world of N ranks with B / N samples each draws exactly the noise one rank draws for B samples.
// per_row % 4 == 0 (one Philox block makes 4 normals); world = 1, rank = 0: element i <- block i / 4 as before.
__global__ __launch_bounds__(256) void randn_kernel(float* out, long n, long per_row, int world, int rank, uint64_t seed, uint64_t stream) {
    const long n4 = (n + 3) / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long e0 = i * 4;
        const long b = e0 / per_row, j = e0 - b * per_row;
        float r[4];
        philox_normal4(seed, stream, (uint64_t)(((b * world + rank) * per_row + j) >> 2), r);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e0 + e < n) out[e0 + e] = r[e];
    }
}
int ew_randn(float* out, long n, uint64_t seed, uint64_t stream, hipStream_t s, long per_row, int world, int rank) {
    if (per_row <= 0 || per_row % 4 || world < 1) { per_row = n > 0 ? ((n + 3) / 4) * 4 : 4; world = 1; rank = 0; }
    int blocks = cdiv_i((n + 3) / 4, 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(randn_kernel, dim3(blocks), dim3(256), 0, s, out, n, per_row, world, rank, seed, stream);
    return 0;
}

// ------------------------------------------------------------------------------------------
// latent reparameterisation + KL (tiny, fp32): last = [mu | logvar] per row
// ------------------------------------------------------------------------------------------
__global__ void latent_fwd_kernel(const float* last, const float* eps, float* z, int B, int Z, double* kl_sum) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < B * Z; i += blockDim.x) {
        const int b = i / Z, d = i - b * Z;
        const float mu = last[b * 2 * Z + d], lv = last[b * 2 * Z + Z + d];
        const float lvc = fminf(fmaxf(lv, -30.f), 30.f);
        const float sd = fminf(fmaxf(expf(0.5f * lvc), 1e-8f), 10.f);
        z[i] = mu + eps[i] * sd;
        acc += 0.5f * (mu * mu + expf(lvc) - lvc - 1.f);
    }
    __shared__ float sm[16];
    const float w = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
        kl_sum[0] = (double)t / B;   // mean over batch (losses.py:32)
    }
}
// dlast = [dz + coef*mu | dz*eps*0.5*std*[in range] + coef*0.5*(e^lv - 1)*[in range]], coef = beta/B
__global__ void latent_bwd_kernel(const float* last, const float* eps, const float* dz, float* dlast, int B, int Z,
                                  float coef) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * Z; i += gridDim.x * blockDim.x) {
        const int b = i / Z, d = i - b * Z;
        const float mu = last[b * 2 * Z + d], lv = last[b * 2 * Z + Z + d];
        const bool inr = (lv >= -30.f) && (lv <= 30.f);
        const float lvc = fminf(fmaxf(lv, -30.f), 30.f);
        const float sd = expf(0.5f * lvc);
        const bool ins = (sd >= 1e-8f) && (sd <= 10.f);
        const float g = dz[i];
        dlast[b * 2 * Z + d] = g + coef * mu;
        float glv = 0.f;
        if (inr) glv = (ins ? g * eps[i] * 0.5f * sd : 0.f) + coef * 0.5f * (expf(lvc) - 1.f);
        dlast[b * 2 * Z + Z + d] = glv;
    }
}
int ew_latent_fwd(const float* last, const float* eps, float* z, int B, int Z, double* kl_sum, hipStream_t s) {
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(1), dim3(256), 0, s, last, eps, z, B, Z, kl_sum);
    return 0;
}
int ew_latent_bwd(const float* last, const float* eps, const float* dz, float* dlast, int B, int Z, float coef,
                  hipStream_t s) {
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(cdiv_i((long)B * Z, 256)), dim3(256), 0, s, last, eps, dz, dlast, B, Z, coef);
    return 0;
}

// ------------------------------------------------------------------------------------------
// decoder stage: posterior/prior combination, kl_2, reparameterisation  (decoder.py:187-212)
//   pz = [mu | lv], qz = [dmu | dlv] fp32 [M][2C]; eps fp32 [M][C]
//   zs_next = dec_out + (mu+dmu) + eps*clamp(exp(.5*clamp(lv+dlv))*std_scale)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stage_fwd_kernel(const float* pz, const float* qz, const float* eps,
                                                       const T* dec_out, long ldd, T* zs_next, long ldz, float* zmap,
                                                       int M, int C, float std_scale, double* kl_part, float inv_b) {
    const long total = (long)M * C;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / C), c = (int)(i - (long)m * C);
        const float mu = pz[(long)m * 2 * C + c], lv = pz[(long)m * 2 * C + C + c];
        const float dmu = qz[(long)m * 2 * C + c], dlv = qz[(long)m * 2 * C + C + c];
        const float lvc = fminf(fmaxf(lv, -30.f), 30.f), dlvc = fminf(fmaxf(dlv, -30.f), 30.f);
        const float var = expf(lvc) + 1e-8f, dvar = expf(dlvc);
        const float df = mu - dmu;
        acc += 0.5f * (dvar / var + df * df / var - dlvc + lvc - 1.f);
        const float lv2 = fminf(fmaxf(lv + dlv, -30.f), 30.f);
        const float sd = fminf(fmaxf(expf(0.5f * lv2) * std_scale, 1e-8f), 10.f);
        const float z = (mu + dmu) + eps[i] * sd;
        if (zmap) zmap[i] = z;
        zs_next[(long)m * ldz + c] = from_f32<T>(to_f32(dec_out[(long)m * ldd + c]) + z);
    }
    __shared__ float sm[4];
    const float w = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) kl_part[blockIdx.x] = (double)(sm[0] + sm[1] + sm[2] + sm[3]) * (double)inv_b;   // summed by ew_rowsum_d
}
// gradients wrt the prior output pz (g_p) and posterior output qz (g_q); coef = beta/B
template <typename T>
__global__ __launch_bounds__(256) void stage_bwd_kernel(const float* pz, const float* qz, const float* eps, const T* dzs,
                                                       long ldd, T* g_p, T* g_q, int M, int C, float coef) {
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / C), c = (int)(i - (long)m * C);
        const float mu = pz[(long)m * 2 * C + c], lv = pz[(long)m * 2 * C + C + c];
        const float dmu = qz[(long)m * 2 * C + c], dlv = qz[(long)m * 2 * C + C + c];
        const bool in1 = (lv >= -30.f) && (lv <= 30.f), in2 = (dlv >= -30.f) && (dlv <= 30.f);
        const float lvc = fminf(fmaxf(lv, -30.f), 30.f), dlvc = fminf(fmaxf(dlv, -30.f), 30.f);
        const float ev = expf(lvc), var = ev + 1e-8f, dvar = expf(dlvc);
        const float df = mu - dmu;
        const float sc = 0.5f * coef;
        const float k_dmu = sc * (-2.f * df / var);
        const float k_mu = -k_dmu;
        const float k_dlv = in2 ? sc * (dvar / var - 1.f) : 0.f;
        const float k_lv = in1 ? sc * (-(dvar + df * df) / (var * var) * ev + 1.f) : 0.f;
        // reparameterisation (training mode: std_scale = 1)
        const float s2 = lv + dlv;
        const bool in3 = (s2 >= -30.f) && (s2 <= 30.f);
        const float sd = expf(0.5f * fminf(fmaxf(s2, -30.f), 30.f));
        const bool ins = (sd >= 1e-8f) && (sd <= 10.f);
        const float g = to_f32(dzs[(long)m * ldd + c]);
        const float g_lv2 = (in3 && ins) ? g * eps[i] * 0.5f * sd : 0.f;
        g_p[(long)m * 2 * C + c] = from_f32<T>(g + k_mu);
        g_p[(long)m * 2 * C + C + c] = from_f32<T>(g_lv2 + k_lv);
        g_q[(long)m * 2 * C + c] = from_f32<T>(g + k_dmu);
        g_q[(long)m * 2 * C + C + c] = from_f32<T>(g_lv2 + k_dlv);
    }
}
int ew_stage_fwd(int dtype, const float* pz, const float* qz, const float* eps, const void* dec_out, long ldd,
                 void* zs_next, long ldz, float* zmap, int M, int C, float std_scale, double* kl_sum, float inv_b,
                 double* kl_part, hipStream_t s) {
    int blocks = cdiv_i((long)M * C, 256);
    if (blocks > 2048) blocks = 2048;
    if (dtype == 1)
        hipLaunchKernelGGL((stage_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, pz, qz, eps, (const bf16_t*)dec_out, ldd,
                           (bf16_t*)zs_next, ldz, zmap, M, C, std_scale, kl_part, inv_b);
    else
        hipLaunchKernelGGL((stage_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, pz, qz, eps, (const float*)dec_out, ldd,
                           (float*)zs_next, ldz, zmap, M, C, std_scale, kl_part, inv_b);
    return ew_rowsum_d(kl_part, blocks, 1, kl_sum, 1.0, s);
}
int ew_stage_bwd(int dtype, const float* pz, const float* qz, const float* eps, const void* dzs, long ldd, void* g_p,
                 void* g_q, int M, int C, float coef, hipStream_t s) {
    int blocks = cdiv_i((long)M * C, 256);
    if (blocks > 4096) blocks = 4096;
    if (dtype == 1)
        hipLaunchKernelGGL((stage_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, pz, qz, eps, (const bf16_t*)dzs, ldd,
                           (bf16_t*)g_p, (bf16_t*)g_q, M, C, coef);
    else
        hipLaunchKernelGGL((stage_bwd_kernel<float>), dim3(blocks), dim3(256), 0, s, pz, qz, eps, (const float*)dzs, ldd,
                           (float*)g_p, (float*)g_q, M, C, coef);
    return 0;
}

// ------------------------------------------------------------------------------------------
// small Linear layers (fp32 master weights, scale = 1/sigma from the spectral-norm pass)
// ------------------------------------------------------------------------------------------
// "head": part[kz][b][o] = scale * sum_{k in slice kz} X[b][k] W[o][k]  (+ bias in slice 0); K large, O small; ew_rowsum adds the slices
template <typename TX>
__global__ __launch_bounds__(256) void linear_head_fwd_kernel(const TX* X, const float* W, const float* bias,
                                                             const float* scale, float* Y, int B, int K, int O) {
    // block = (output o, K slice kz); a thread's W vector meets 8 batch rows at a time, all loads of a round independent
    constexpr int BG = 8;
    const int o = blockIdx.x;
    const int ks = gridDim.y, kz = blockIdx.y;
    const int nv = K / 8;
    const int v_lo = (int)((long)nv * kz / ks), v_hi = (int)((long)nv * (kz + 1) / ks);
    const float sc = scale ? *scale : 1.f;
    __shared__ float sm[4][BG];
    for (int b0 = 0; b0 < B; b0 += BG) {
        float acc[BG];
#pragma unroll
        for (int bb = 0; bb < BG; ++bb) acc[bb] = 0.f;
        for (int v = v_lo + threadIdx.x; v < v_hi; v += 256) {
            float w[8];
            load8(W + (long)o * K + v * 8, w);
#pragma unroll
            for (int bb = 0; bb < BG; ++bb) {
                if (b0 + bb < B) {
                    float x[8];
                    load8(X + (long)(b0 + bb) * K + v * 8, x);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[bb] += w[e] * x[e];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < BG; ++bb) {
            const float wsum = wave_sum(acc[bb]);
            if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][bb] = wsum;
        }
        __syncthreads();
        if (threadIdx.x < BG && b0 + (int)threadIdx.x < B) {
            float t = (sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]) * sc;
            if (kz == 0 && bias) t += bias[o];
            Y[((long)kz * B + (b0 + threadIdx.x)) * O + o] = t;
        }
    }
}
// dX[b][k] = scale * sum_o dY[b][o] W[o][k] (+ addend[b][k]); output TX
template <typename TX>
__global__ __launch_bounds__(256) void linear_head_bwd_dx_kernel(const float* dY, const float* W, const float* scale,
                                                                const TX* addend, TX* dX, int B, int K, int O) {
    const int nv = K / 8;
    const float sc = scale ? *scale : 1.f;
    const long total = (long)B * nv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / nv), v = (int)(i - (long)b * nv);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < O; ++o) {
            const float g = dY[(long)b * O + o];
            float w[8];
            load8(W + (long)o * K + v * 8, w);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += g * w[e];
        }
        float a[8];
        if (addend) load8(addend + (long)b * K + v * 8, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = acc[e] * sc + (addend ? a[e] : 0.f);
        store8(dX + (long)b * K + v * 8, acc);
    }
}
// dW[o][k] = sum_b dY[b][o] X[b][k]  (plain store); db[o] = sum_b dY[b][o]
template <typename TX>
__global__ __launch_bounds__(256) void linear_head_bwd_dw_kernel(const float* dY, const TX* X, float* dW, float* db, int B,
                                                                int K, int O) {
    const int nv = K / 8;
    const long total = (long)O * nv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i / nv), v = (int)(i - (long)o * nv);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            const float g = dY[(long)b * O + o];
            float x[8];
            load8(X + (long)b * K + v * 8, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += g * x[e];
        }
        store8(dW + (long)o * K + v * 8, acc);
    }
    if (db && blockIdx.x == 0) {
        for (int o = threadIdx.x; o < O; o += 256) {
            float t = 0.f;
            for (int b = 0; b < B; ++b) t += dY[(long)b * O + o];
            db[o] = t;
        }
    }
}
int ew_linear_head_fwd(int xdtype, const void* X, const float* W, const float* bias, const float* scale, float* Y, int B,
                       int K, int O, float* part, hipStream_t s) {
    int ks = cdiv_i(K / 8, 512);       // two vectors per thread: the pass is latency-bound, not bandwidth-bound
    if (ks < 1) ks = 1;
    if (ks > 128) ks = 128;
    dim3 grid(O, ks);
    if (xdtype == 1) hipLaunchKernelGGL((linear_head_fwd_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)X, W, bias, scale, part, B, K, O);
    else hipLaunchKernelGGL((linear_head_fwd_kernel<float>), grid, dim3(256), 0, s, (const float*)X, W, bias, scale, part, B, K, O);
    return ew_rowsum(part, 1, ks, B * O, Y, nullptr, 1.0, s);
}
int ew_linear_head_bwd(int xdtype, const float* dY, const void* X, const float* W, const float* scale, const void* addend,
                       void* dX, float* dW, float* db, int B, int K, int O, hipStream_t s) {
    int blocks = cdiv_i((long)B * (K / 8), 256);
    if (blocks > 4096) blocks = 4096;
    int blocks2 = cdiv_i((long)O * (K / 8), 256);
    if (blocks2 > 4096) blocks2 = 4096;
    if (xdtype == 1) {
        if (dX) hipLaunchKernelGGL((linear_head_bwd_dx_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, dY, W, scale, (const bf16_t*)addend, (bf16_t*)dX, B, K, O);
        if (dW) hipLaunchKernelGGL((linear_head_bwd_dw_kernel<bf16_t>), dim3(blocks2), dim3(256), 0, s, dY, (const bf16_t*)X, dW, db, B, K, O);
    } else {
        if (dX) hipLaunchKernelGGL((linear_head_bwd_dx_kernel<float>), dim3(blocks), dim3(256), 0, s, dY, W, scale, (const float*)addend, (float*)dX, B, K, O);
        if (dW) hipLaunchKernelGGL((linear_head_bwd_dw_kernel<float>), dim3(blocks2), dim3(256), 0, s, dY, (const float*)X, dW, db, B, K, O);
    }
    return 0;
}

// "expand": Y[b][o] = scale * sum_k X[b][k] W[o][k] + bias[o]; K tiny (<= 64), O = T*C large; Y in T
template <typename T>
__global__ __launch_bounds__(256) void linear_expand_fwd_kernel(const float* X, const float* W, const float* bias,
                                                               const float* scale, T* Y, int B, int K, int O) {
    const float sc = scale ? *scale : 1.f;
    const long total = (long)B * O;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / O), o = (int)(i - (long)b * O);
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += X[b * K + k] * W[(long)o * K + k];
        Y[i] = from_f32<T>(acc * sc + bias[o]);
    }
}
// dW[o][k] = sum_b dY[b][o] X[b][k]; db[o] = sum_b dY[b][o]
template <typename T>
__global__ __launch_bounds__(256) void linear_expand_bwd_dw_kernel(const T* dY, const float* X, float* dW, float* db, int B,
                                                                  int K, int O) {
    const long total = (long)O * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i / K), k = (int)(i - (long)o * K);
        float acc = 0.f, accb = 0.f;
        for (int b = 0; b < B; ++b) {
            const float g = to_f32(dY[(long)b * O + o]);
            acc += g * X[b * K + k];
            accb += g;
        }
        dW[i] = acc;
        if (k == 0) db[o] = accb;
    }
}
// dX[b][k] = scale * sum_o dY[b][o] W[o][k]   (one block per (b,k))
template <typename T>
__global__ __launch_bounds__(256) void linear_expand_bwd_dx_kernel(const T* dY, const float* W, const float* scale,
                                                                  float* dX, int B, int K, int O) {
    const int b = blockIdx.x / K, k = blockIdx.x - b * K;
    float acc = 0.f;
    for (int o = threadIdx.x; o < O; o += 256) acc += to_f32(dY[(long)b * O + o]) * W[(long)o * K + k];
    __shared__ float sm[4];
    const float w = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) dX[b * K + k] = (sm[0] + sm[1] + sm[2] + sm[3]) * (scale ? *scale : 1.f);
}
int ew_linear_expand_fwd(int dtype, const float* X, const float* W, const float* bias, const float* scale, void* Y, int B,
                         int K, int O, hipStream_t s) {
    int blocks = cdiv_i((long)B * O, 256);
    if (dtype == 1) hipLaunchKernelGGL((linear_expand_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, X, W, bias, scale, (bf16_t*)Y, B, K, O);
    else hipLaunchKernelGGL((linear_expand_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, X, W, bias, scale, (float*)Y, B, K, O);
    return 0;
}
int ew_linear_expand_bwd(int dtype, const void* dY, const float* X, const float* W, const float* scale, float* dX, float* dW,
                         float* db, int B, int K, int O, hipStream_t s) {
    int blocks = cdiv_i((long)O * K, 256);
    if (dtype == 1) {
        hipLaunchKernelGGL((linear_expand_bwd_dw_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)dY, X, dW, db, B, K, O);
        if (dX) hipLaunchKernelGGL((linear_expand_bwd_dx_kernel<bf16_t>), dim3(B * K), dim3(256), 0, s, (const bf16_t*)dY, W, scale, dX, B, K, O);
    } else {
        hipLaunchKernelGGL((linear_expand_bwd_dw_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)dY, X, dW, db, B, K, O);
        if (dX) hipLaunchKernelGGL((linear_expand_bwd_dx_kernel<float>), dim3(B * K), dim3(256), 0, s, (const float*)dY, W, scale, dX, B, K, O);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// augmentation + collate on the HBM-resident dataset (internal layout [P][T][N], dtype T)
//   out[b] = lam*(scale*(x[idx] + 0.05*noise)) + (1-lam)*x[mix]     (augmentation.py:58-124 order)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void augment_kernel(const T* data, T* out, long sample_elems, const int* idx,
                                                     const unsigned long long* noise_seed, const float* scale,
                                                     const int* mix_idx, const float* lam) {
    const int b = blockIdx.y;
    const T* src = data + (long)idx[b] * sample_elems;
    const int mi = mix_idx[b];
    const T* oth = mi >= 0 ? data + (long)mi * sample_elems : nullptr;
    const unsigned long long ns = noise_seed[b];
    const float sc = scale[b], lm = lam[b];
    T* dst = out + (long)b * sample_elems;
    const long nv = sample_elems / 8;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) {
        float x[8];
        load8(src + v * 8, x);
        if (ns) {
            float n0[4], n1[4];
            philox_normal4(ns, 0x41554721ull, (uint64_t)(v * 2), n0);
            philox_normal4(ns, 0x41554721ull, (uint64_t)(v * 2 + 1), n1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] += 0.05f * n0[e]; x[e + 4] += 0.05f * n1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] *= sc;
        if (oth) {
            float o[8];
            load8(oth + v * 8, o);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = lm * x[e] + (1.f - lm) * o[e];
        }
        store8(dst + v * 8, x);
    }
}
int ew_augment(int dtype, const void* data, void* out, long sample_elems, int batch, const int* idx,
               const unsigned long long* noise_seed, const float* scale, const int* mix_idx, const float* lam, hipStream_t s) {
    int bx = cdiv_i(sample_elems / 8, 256);
    if (bx > 512) bx = 512;
    dim3 grid(bx, batch);
    if (dtype == 1) hipLaunchKernelGGL((augment_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)data, (bf16_t*)out, sample_elems, idx, noise_seed, scale, mix_idx, lam);
    else hipLaunchKernelGGL((augment_kernel<float>), grid, dim3(256), 0, s, (const float*)data, (float*)out, sample_elems, idx, noise_seed, scale, mix_idx, lam);
    return 0;
}

// y[i] += a * x[i]  (fp32, small arrays)
__global__ void axpy_kernel(float* y, const float* x, float a, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}
// d0 = a*x[0..n), d1 = a*x[n..2n), d2 = a*x[2n..3n)   (recon head: unit-weight affine / bias gradients -> weighted, written not added)
__global__ void scale3_kernel(float* d0, float* d1, float* d2, const float* x, float a, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 3 * n; i += (long)gridDim.x * 256) {
        const float v = a * x[i];
        if (i < n) d0[i] = v; else if (i < 2 * n) d1[i - n] = v; else d2[i - 2 * n] = v;
    }
}
int ew_scale3(float* d0, float* d1, float* d2, const float* x, float a, long n, hipStream_t s) {
    int blocks = cdiv_i(3 * n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale3_kernel, dim3(blocks), dim3(256), 0, s, d0, d1, d2, x, a, n);
    return 0;
}
__global__ __launch_bounds__(256) void fill_from_scalar_kernel(float* dst, const float* src, long n) {
    const float v = *src;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = v;
}
int ew_fill_from_scalar(float* dst, const float* src_scalar, long n, hipStream_t s) {
    if (n <= 0) return 0;
    const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(fill_from_scalar_kernel, dim3(blocks), dim3(256), 0, s, dst, src_scalar, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// gradient payload of the data-parallel step: fp32 arena range <-> bf16 wire copy (round to nearest even), 8 elements per thread
template <bool VEC>
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long step = (long)gridDim.x * 256 * 8;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += step) {
        if (VEC && i + 8 <= n) {
            const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            store8(dst + i, v);
        } else {
            for (long j = i; j < i + 8 && j < n; ++j) dst[j] = from_f32<bf16_t>(src[j]);
        }
    }
}
template <bool VEC>
__global__ __launch_bounds__(256) void unpack_bf16_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
    const long step = (long)gridDim.x * 256 * 8;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += step) {
        if (VEC && i + 8 <= n) {
            float v[8];
            load8(src + i, v);
            *reinterpret_cast<float4*>(dst + i) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            for (long j = i; j < i + 8 && j < n; ++j) dst[j] = to_f32(src[j]);
        }
    }
}
int ew_pack_bf16(const float* src, void* dst, long n, hipStream_t s) {
    if (n <= 0) return 0;
    int blocks = (int)std::min<long>((n + 2047) / 2048, 4096);
    const bool vec = ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    if (vec) hipLaunchKernelGGL(pack_bf16_kernel<true>, dim3(blocks), dim3(256), 0, s, src, (bf16_t*)dst, n);
    else hipLaunchKernelGGL(pack_bf16_kernel<false>, dim3(blocks), dim3(256), 0, s, src, (bf16_t*)dst, n);
    return 0;
}
int ew_unpack_bf16(const void* src, float* dst, long n, hipStream_t s) {
    if (n <= 0) return 0;
    int blocks = (int)std::min<long>((n + 2047) / 2048, 4096);
    const bool vec = ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    if (vec) hipLaunchKernelGGL(unpack_bf16_kernel<true>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, dst, n);
    else hipLaunchKernelGGL(unpack_bf16_kernel<false>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, dst, n);
    return 0;
}
int ew_axpy(float* y, const float* x, float a, long n, hipStream_t s) {
    int blocks = cdiv_i(n, 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, s, y, x, a, n);
    return 0;
}
__global__ void scale_kernel(float* y, float a, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] *= a;
}
int ew_scale(float* y, float a, long n, hipStream_t s) {
    int blocks = cdiv_i(n, 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, s, y, a, n);
    return 0;
}
// cast fp32 [rows][C] -> T [rows][ld]
template <typename T>
__global__ void cast_rows_kernel(const float* src, long lds_, T* dst, long ldd, int rows, int C) {
    const long total = (long)rows * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / C), c = (int)(i - (long)r * C);
        dst[(long)r * ldd + c] = from_f32<T>(src[(long)r * lds_ + c]);
    }
}
int ew_cast_rows(int dtype, const float* src, long lds_, void* dst, long ldd, int rows, int C, hipStream_t s) {
    int blocks = cdiv_i((long)rows * C, 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (dtype == 1) hipLaunchKernelGGL((cast_rows_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, src, lds_, (bf16_t*)dst, ldd, rows, C);
    else hipLaunchKernelGGL((cast_rows_kernel<float>), dim3(blocks), dim3(256), 0, s, src, lds_, (float*)dst, ldd, rows, C);
    return 0;
}

// ------------------------------------------------------------------------------------------
// Input pipeline: per-node MinMaxScaler(-0.7, 0.7) as the reference fits it on a row sample
// (modules/data_preprocess.py:65-165 -> sklearn.preprocessing.MinMaxScaler) and the scaled copy of the raw
// [P][T][N] array straight into the engine's resident layout [P][T][N] (the reference's transpose to [P,N,T],
// SimulGen-VAE.py:282, and ours back cancel).  All HBM-bound streaming passes.
// ------------------------------------------------------------------------------------------
// partial[rs][2][N]: min / max over rows rs, rs+RS, ... of the column; NaNs are ignored like np.nanmin/nanmax
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* rows, long n_rows, int N, float* partial) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= N) return;
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, INFINITY), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (long r = blockIdx.y; r < n_rows; r += gridDim.y) {
        const float4 v = *reinterpret_cast<const float4*>(rows + r * N + c);
        lo.x = fminf(lo.x, v.x); lo.y = fminf(lo.y, v.y); lo.z = fminf(lo.z, v.z); lo.w = fminf(lo.w, v.w);
        hi.x = fmaxf(hi.x, v.x); hi.y = fmaxf(hi.y, v.y); hi.z = fmaxf(hi.z, v.z); hi.w = fmaxf(hi.w, v.w);
    }
    float* p = partial + (long)blockIdx.y * 2 * N;
    *reinterpret_cast<float4*>(p + c) = lo;
    *reinterpret_cast<float4*>(p + N + c) = hi;
}
__global__ __launch_bounds__(256) void minmax_final_kernel(const float* partial, int RS, int N, float* mn, float* mx, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float lo = accumulate ? mn[c] : INFINITY, hi = accumulate ? mx[c] : -INFINITY;
    for (int r = 0; r < RS; ++r) { lo = fminf(lo, partial[(long)r * 2 * N + c]); hi = fmaxf(hi, partial[(long)r * 2 * N + N + c]); }
    mn[c] = lo; mx[c] = hi;
}
// sklearn MinMaxScaler._partial_fit: scale_ = (hi-lo)/handle_zeros(data_max-data_min), min_ = lo - data_min*scale_
__global__ __launch_bounds__(256) void minmax_coeffs_kernel(const float* mn, const float* mx, int N, float lo, float hi, float* scale, float* offset) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float range = mx[c] - mn[c];
    if (range < 10.f * 1.1920929e-07f) range = 1.f;          // _handle_zeros_in_scale (10 * eps of the dtype)
    const float s = (hi - lo) / range;
    scale[c] = s;
    offset[c] = lo - mn[c] * s;
}
template <typename T>
__global__ __launch_bounds__(256) void scale_convert_kernel(const float* src, const float* scale, const float* offset, T* dst,
                                                           long n_rows, int N) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (c >= N) return;
    float s[8], o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = scale[c + e]; o[e] = offset[c + e]; }
    for (long r = blockIdx.y; r < n_rows; r += gridDim.y) {
        float v[8];
        load8(src + r * N + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * s[e] + o[e];     // sklearn: X *= scale_; X += min_
        store8(dst + r * N + c, v);
    }
}
int ew_minmax_fit(const float* rows, long n_rows, int N, float* mn, float* mx, float* partial, int RS, int accumulate, hipStream_t s) {
    if (n_rows <= 0 || N % 4) return -1;
    if (RS > n_rows) RS = (int)n_rows;
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(cdiv_i(N / 4, 256), RS), dim3(256), 0, s, rows, n_rows, N, partial);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(cdiv_i(N, 256)), dim3(256), 0, s, partial, RS, N, mn, mx, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int ew_minmax_coeffs(const float* mn, const float* mx, int N, float lo, float hi, float* scale, float* offset, hipStream_t s) {
    hipLaunchKernelGGL(minmax_coeffs_kernel, dim3(cdiv_i(N, 256)), dim3(256), 0, s, mn, mx, N, lo, hi, scale, offset);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int ew_scale_convert(int dtype, const float* src, const float* scale, const float* offset, void* dst, long n_rows, int N, hipStream_t s) {
    if (n_rows <= 0) return 0;
    if (N % 8) return -1;
    int ry = (int)(n_rows < 2048 ? n_rows : 2048);
    dim3 grid(cdiv_i(N / 8, 256), ry);
    if (dtype == 1) hipLaunchKernelGGL((scale_convert_kernel<bf16_t>), grid, dim3(256), 0, s, src, scale, offset, (bf16_t*)dst, n_rows, N);
    else hipLaunchKernelGGL((scale_convert_kernel<float>), grid, dim3(256), 0, s, src, scale, offset, (float*)dst, n_rows, N);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
