// Multi-tensor HBM-bound passes over the fp32 master weights (gfx950):
//  * legacy spectral-norm power iteration (torch nn/utils/spectral_norm.py compute_weight, applied by
//    reference modules/common.py:15-37): v <- norm(W^T u), u <- norm(W v), sigma = u.(W v)
//  * <G, W> per weight for the spectral-norm chain rule, fused AdamW (torch.optim.AdamW defaults,
//    reference modules/train.py:92,168) + gradient 2-norm (modules/train.py:156-161)
//  * compute-dtype weight copies in both GEMM layouts.
// Every pass is ONE launch over a work-item table (desc, chunk) covering all tensors.
// Weights are [taps][rows][cols] fp32 (cols contiguous); u is [rows], v is [taps*cols].
#include "sgv_ew.h"

// ---- pass 1: tpart[row block][tap][c] = sum_{r in block} W[tap][r][c] * u[r] over a 64-row x 1024-col block ----------
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const SNDesc* descs, const WorkItem* items) {
    const WorkItem it = items[blockIdx.x];
    const SNDesc d = descs[it.desc];
    const int cb = (d.cols + SN_COLS_PER_ITEM - 1) / SN_COLS_PER_ITEM;
    const int rb = (d.rows + SN_ROWS_PER_ITEM - 1) / SN_ROWS_PER_ITEM;
    const int tap = it.chunk / (rb * cb);
    const int rem = it.chunk - tap * rb * cb;
    const int r_lo = (rem / cb) * SN_ROWS_PER_ITEM, c = (rem % cb) * SN_COLS_PER_ITEM + threadIdx.x * 4;
    if (c >= d.cols) return;
    const int r_hi = min(d.rows, r_lo + SN_ROWS_PER_ITEM);
    const float* W = d.W + ((long)tap * d.rows) * d.cols + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int r = r_lo; r < r_hi; ++r) {
        const float4 w = *reinterpret_cast<const float4*>(W + (long)r * d.cols);
        const float ur = d.u[r];
        a0 += w.x * ur; a1 += w.y * ur; a2 += w.z * ur; a3 += w.w * ur;
    }
    float* t = d.tpart + ((long)(r_lo / SN_ROWS_PER_ITEM) * d.taps + tap) * d.cols + c;
    *reinterpret_cast<float4*>(t) = make_float4(a0, a1, a2, a3);
}
// tmp_t[i] = sum over the row blocks of tpart[block][i] in a fixed order; item = (desc, chunk of 64 elements of taps*cols);
// 16 lanes x 16 float4 columns: lane l sums blocks l, l+16, ..., then the lanes are added in index order
__global__ __launch_bounds__(256) void sn_tsum_kernel(const SNDesc* descs, const WorkItem* items) {
    __shared__ float4 sm[16][17];
    const WorkItem it = items[blockIdx.x];
    const SNDesc d = descs[it.desc];
    const long n = (long)d.taps * d.cols;
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const long i = (long)it.chunk * 64 + cq * 4;
    const int rb = (d.rows + SN_ROWS_PER_ITEM - 1) / SN_ROWS_PER_ITEM;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        // four independent chains (the recon head has 1485 row blocks: 93 dependent adds per lane otherwise, one load in flight each)
        float4 a1 = a, a2 = a, a3 = a;
        int r = rl;
        for (; r + 48 < rb; r += 64) {
            const float4 v0 = *reinterpret_cast<const float4*>(d.tpart + (long)r * n + i);
            const float4 v1 = *reinterpret_cast<const float4*>(d.tpart + (long)(r + 16) * n + i);
            const float4 v2 = *reinterpret_cast<const float4*>(d.tpart + (long)(r + 32) * n + i);
            const float4 v3 = *reinterpret_cast<const float4*>(d.tpart + (long)(r + 48) * n + i);
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
            a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; r < rb; r += 16) {
            const float4 v = *reinterpret_cast<const float4*>(d.tpart + (long)r * n + i);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        a.x = (a.x + a1.x) + (a2.x + a3.x); a.y = (a.y + a1.y) + (a2.y + a3.y);
        a.z = (a.z + a1.z) + (a2.z + a3.z); a.w = (a.w + a1.w) + (a2.w + a3.w);
    }
    sm[rl][cq] = a;
    __syncthreads();
    if (rl == 0 && i < n) {
        float4 t = sm[0][cq];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = sm[k][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(d.tmp_t + i) = t;
    }
}
// tmp_s[r] = sum over the (tap, column block) partials in a fixed order; item = (desc, chunk of 64 rows): 4 lanes x 64 rows,
// lane l sums partials l, l+4, ..., then the lanes are added in index order
__global__ __launch_bounds__(256) void sn_ssum_kernel(const SNDesc* descs, const WorkItem* items) {
    __shared__ float sm[4][64];
    const WorkItem it = items[blockIdx.x];
    const SNDesc d = descs[it.desc];
    const int np = d.taps * ((d.cols + SN_COLS_PER_ITEM - 1) / SN_COLS_PER_ITEM);
    const int rq = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int r = it.chunk * 64 + rq;
    float a = 0.f;
    if (r < d.rows)
        for (int k = kl; k < np; k += 4) a += d.spart[(long)k * d.rows + r];
    sm[kl][rq] = a;
    __syncthreads();
    if (kl == 0 && r < d.rows) d.tmp_s[r] = ((sm[0][rq] + sm[1][rq]) + sm[2][rq]) + sm[3][rq];
}

// ---- pass 2: v = t / max(||t||, 1e-12) ------------------------------------------------------------
// sum of squares / dot of one vector by one 1024-thread block: 16-byte loads when the base is 16-byte aligned, float
// partials per thread (<= ~100 terms), fp64 across threads
__device__ __forceinline__ double block1024_dot(const float* a, const float* b, long n) {
    __shared__ double smd[16];
    float acc = 0.f;
    const bool vec = ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0;
    const long n4 = vec ? n >> 2 : 0;
    for (long i = threadIdx.x; i < n4; i += 1024) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 1024) acc += a[i] * b[i];
    const double w = wave_sum_d((double)acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smd[threadIdx.x >> 6] = w;
    __syncthreads();
    double tot = 0.0;
    for (int i = 0; i < 16; ++i) tot += smd[i];
    return tot;
}
__device__ __forceinline__ void block1024_scale(const float* src, float* dst, float f, long n) {
    const bool vec = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    const long n4 = vec ? n >> 2 : 0;
    for (long i = threadIdx.x; i < n4; i += 1024) {
        float4 x = reinterpret_cast<const float4*>(src)[i];
        x.x *= f; x.y *= f; x.z *= f; x.w *= f;
        reinterpret_cast<float4*>(dst)[i] = x;
    }
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 1024) dst[i] = src[i] * f;
}
__global__ __launch_bounds__(1024) void sn_norm_v_kernel(const SNDesc* descs, int ndesc) {
    const SNDesc d = descs[blockIdx.x];
    if (!d.active) return;
    const long n = (long)d.taps * d.cols;
    const double tot = block1024_dot(d.tmp_t, d.tmp_t, n);
    const float inv = 1.0f / fmaxf((float)sqrt(tot), 1e-12f);
    block1024_scale(d.tmp_t, d.v, inv, n);
}

// ---- pass 3: spart[(tap, col block)][r] = sum_{c in block} W[tap][r][c] * v[tap][c]; one wave per row ---------------------------
__global__ __launch_bounds__(256) void sn_w_v_kernel(const SNDesc* descs, const WorkItem* items) {
    const WorkItem it = items[blockIdx.x];
    const SNDesc d = descs[it.desc];
    const int cb = (d.cols + SN_COLS_PER_ITEM - 1) / SN_COLS_PER_ITEM;
    const int rb = (d.rows + SN_ROWS_PER_ITEM - 1) / SN_ROWS_PER_ITEM;
    const int tap = it.chunk / (rb * cb);
    const int rem = it.chunk - tap * rb * cb;
    const int r_lo = (rem / cb) * SN_ROWS_PER_ITEM;
    const int c_lo = (rem % cb) * SN_COLS_PER_ITEM;
    const int c_hi = min(d.cols, c_lo + SN_COLS_PER_ITEM);
    const int r_hi = min(d.rows, r_lo + SN_ROWS_PER_ITEM);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* vv = d.v + (long)tap * d.cols;
    float* sp = d.spart + ((long)tap * cb + (rem % cb)) * d.rows;
    if (d.wc) {   // bf16 engine: the GEMMs multiply by this copy, and it is half the bytes of the master
        const unsigned short* Wc = reinterpret_cast<const unsigned short*>(d.wc);
        for (int r = r_lo + wave; r < r_hi; r += 4) {
            const unsigned short* W = Wc + ((long)tap * d.rows + r) * d.cols;
            float acc = 0.f;
            for (int c = c_lo + lane * 8; c < c_hi; c += 512) {
                const uint4 q = *reinterpret_cast<const uint4*>(W + c);
                const float4 x0 = *reinterpret_cast<const float4*>(vv + c), x1 = *reinterpret_cast<const float4*>(vv + c + 4);
                acc += __uint_as_float(q.x << 16) * x0.x + __uint_as_float(q.x & 0xffff0000u) * x0.y
                     + __uint_as_float(q.y << 16) * x0.z + __uint_as_float(q.y & 0xffff0000u) * x0.w
                     + __uint_as_float(q.z << 16) * x1.x + __uint_as_float(q.z & 0xffff0000u) * x1.y
                     + __uint_as_float(q.w << 16) * x1.z + __uint_as_float(q.w & 0xffff0000u) * x1.w;
            }
            acc = wave_sum(acc);
            if (lane == 0) sp[r] = acc;
        }
        return;
    }
    for (int r = r_lo + wave; r < r_hi; r += 4) {
        const float* W = d.W + ((long)tap * d.rows + r) * d.cols;
        float acc = 0.f;
        for (int c = c_lo + lane * 4; c < c_hi; c += 256) {
            const float4 w = *reinterpret_cast<const float4*>(W + c);
            const float4 x = *reinterpret_cast<const float4*>(vv + c);
            acc += w.x * x.x + w.y * x.y + w.z * x.z + w.w * x.w;
        }
        acc = wave_sum(acc);
        if (lane == 0) sp[r] = acc;
    }
}

// ---- pass 4: u = s / max(||s||, 1e-12) (train), sigma = u . s ----------------------------------------
__global__ __launch_bounds__(1024) void sn_norm_u_kernel(const SNDesc* descs, int train) {
    const SNDesc d = descs[blockIdx.x];
    if (!d.active) return;
    const double tot = train ? block1024_dot(d.tmp_s, d.tmp_s, d.rows) : block1024_dot(d.u, d.tmp_s, d.rows);
    float sigma;
    if (train) {
        const float nrm = fmaxf((float)sqrt(tot), 1e-12f);
        block1024_scale(d.tmp_s, d.u, 1.0f / nrm, d.rows);
        sigma = (float)(tot / (double)nrm);   // u . s with u = s/nrm
    } else {
        sigma = (float)tot;
    }
    if (threadIdx.x == 0) { d.sigma[0] = sigma; d.sigma[1] = 1.0f / sigma; }
}

int opt_sn_power_iteration(const SNDesc* descs_dev, const WorkItem* items1, int n1, const WorkItem* items3, int n3,
                           const WorkItem* items_ts, int n_ts, const WorkItem* items_ss, int n_ss, int ndesc, int train, hipStream_t s) {
    if (train) {
        if (n1 > 0) hipLaunchKernelGGL(sn_wt_u_kernel, dim3(n1), dim3(256), 0, s, descs_dev, items1);
        if (n_ts > 0) hipLaunchKernelGGL(sn_tsum_kernel, dim3(n_ts), dim3(256), 0, s, descs_dev, items_ts);
        hipLaunchKernelGGL(sn_norm_v_kernel, dim3(ndesc), dim3(1024), 0, s, descs_dev, ndesc);
    }
    if (n3 > 0) hipLaunchKernelGGL(sn_w_v_kernel, dim3(n3), dim3(256), 0, s, descs_dev, items3);
    if (n_ss > 0) hipLaunchKernelGGL(sn_ssum_kernel, dim3(n_ss), dim3(256), 0, s, descs_dev, items_ss);
    hipLaunchKernelGGL(sn_norm_u_kernel, dim3(ndesc), dim3(1024), 0, s, descs_dev, train);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- <G, W> per spectrally-normalised weight ---------------------------------------------------------
__global__ __launch_bounds__(256) void sn_grad_dot_kernel(const SNDesc* descs, const WorkItem* items, float* dot_part) {
    const WorkItem it = items[blockIdx.x];
    const SNDesc d = descs[it.desc];
    const long n = (long)d.taps * d.rows * d.cols;
    const long lo = (long)it.chunk * OPT_CHUNK;
    const long hi = min(n, lo + OPT_CHUNK);
    float acc = 0.f;
    for (long i = lo + threadIdx.x * 4; i < hi; i += 1024) {
        const float4 g = *reinterpret_cast<const float4*>(d.G + i);
        const float4 w = *reinterpret_cast<const float4*>(d.W + i);
        acc += g.x * w.x + g.y * w.y + g.z * w.z + g.w * w.w;
    }
    __shared__ float sm[4];
    const float w = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = w;
    __syncthreads();
    // <G, W_eff> = <G, W> / sigma.  Only the small Linear layers take this weight-sized route; conv layers get
    // the same number as sum dY*(y - bias) inside their dY-producing kernels (ew.hip).
    if (threadIdx.x == 0) dot_part[blockIdx.x] = (sm[0] + sm[1] + sm[2] + sm[3]) * d.sigma[1];
}
int opt_sn_grad_dot(const SNDesc* descs_dev, const WorkItem* items, int n, float* dot_part, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(sn_grad_dot_kernel, dim3(n), dim3(256), 0, s, descs_dev, items, dot_part);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- fused AdamW (+ spectral-norm chain rule, + grad norm) -------------------------------------------
// g_orig = (G - <G,W_eff> u v^T) / sigma with <G,W_eff> = dot / sigma.
template <bool UPDATE, bool WC_BF16>
__global__ __launch_bounds__(256) void adamw_kernel(const AdamDesc* adam, const SNDesc* sn, const WorkItem* items, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2sqrt,
                                                   double* gnorm_sq, const float* gscale) {
    const WorkItem it = items[blockIdx.x];
    const AdamDesc a = adam[it.desc];
    const long lo = (long)it.chunk * OPT_CHUNK;
    const long hi = min(a.n, lo + OPT_CHUNK);
    float inv_sigma = 1.f, cdot = 0.f;
    const float* u = nullptr;
    const float* v = nullptr;
    if (a.sn >= 0) {
        const SNDesc d = sn[a.sn];
        inv_sigma = d.sigma[1];
#pragma unroll
        for (int k = 0; k < SGV_DOT_SLOTS; ++k) cdot += d.dot[k];
        u = d.u; v = d.v;
    }
    const long rc = (long)a.rows * a.cols;
    float nacc = 0.f;
    const float step = lr / bc1;
    for (long i = lo + threadIdx.x * 4; i < hi; i += 1024) {
        float4 g = *reinterpret_cast<const float4*>(a.g + i);
        if (a.sn >= 0) {
            const long tap = i / rc;
            const long rem = i - tap * rc;
            const int r = (int)(rem / a.cols), c = (int)(rem - (long)r * a.cols);
            const float ur = u[r] * cdot;
            const float4 vv = *reinterpret_cast<const float4*>(v + tap * a.cols + c);
            g.x = (g.x - ur * vv.x) * inv_sigma; g.y = (g.y - ur * vv.y) * inv_sigma;
            g.z = (g.z - ur * vv.z) * inv_sigma; g.w = (g.w - ur * vv.w) * inv_sigma;
        }
        nacc += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
        if constexpr (UPDATE) {
            if (gscale) { const float gs = gscale[0]; g.x *= gs; g.y *= gs; g.z *= gs; g.w *= gs; }   // clip_grad_norm_ coefficient
            float4 p = *reinterpret_cast<const float4*>(a.p + i);
            float4 m = *reinterpret_cast<const float4*>(a.m + i);
            float4 vs = *reinterpret_cast<const float4*>(a.v + i);
            const float decay = 1.f - lr * wd;
#define SGV_ADAM1(F)                                                    \
            p.F *= decay;                                               \
            m.F = m.F * b1 + (1.f - b1) * g.F;                          \
            vs.F = vs.F * b2 + (1.f - b2) * g.F * g.F;                  \
            p.F -= step * (m.F / (sqrtf(vs.F) / bc2sqrt + eps));
            SGV_ADAM1(x) SGV_ADAM1(y) SGV_ADAM1(z) SGV_ADAM1(w)
#undef SGV_ADAM1
            *reinterpret_cast<float4*>(a.p + i) = p;
            *reinterpret_cast<float4*>(a.m + i) = m;
            *reinterpret_cast<float4*>(a.v + i) = vs;
            if (WC_BF16 && a.wc) {   // bf16 compute copy written in the same pass (saves re-reading the master)
                typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                bf16x4_t o;
                o[0] = (__bf16)p.x; o[1] = (__bf16)p.y; o[2] = (__bf16)p.z; o[3] = (__bf16)p.w;
                *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(a.wc) + i) = o;
            }
        }
    }
    __shared__ float sm[4];
    const float w = wave_sum(nacc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) gnorm_sq[blockIdx.x] = (double)(sm[0] + sm[1] + sm[2] + sm[3]);      // per-item partial, summed in index order later
}
// ---- AdamW over 64x64 tiles of the spectrally-normalised conv weights ------------------------------
// Same update as adamw_kernel, plus everything else that needs the freshly updated weight while it is in
// registers: the compute-dtype copy wc, the transposed/tap-flipped copy wct (through an LDS tile) and the
// first half of the NEXT forward's power iteration, tpart[row tile][tap][c] = sum_{r in tile} W_new[r][c] * u[r]  (u is only
// modified by the forward itself).  Saves two full passes over the 1.6 GB of master weights per step.
// GLP: the gradient is read from the bf16 data-parallel wire copy (g_lp, same element offsets as the fp32 arena g_base) instead of
// the arena -- the averaged bucket goes straight from the collective into the update, no unpack pass in between.
// NTM: cache policy of the once-per-step streams.  bit 0: moments m / v (read once, written once, next touched a step later),
// bit 1: the fp32 gradient read, bit 2: the fp32 master weight, bit 3: the bf16 operand copies (read by next step's GEMMs, 1.6 GB
// later) -- non-temporal, so that 12.6 GB per step do not sweep the operand
// panels of the GEMMs running beside this pass out of L2 / Infinity Cache.
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ld4(const float* p) {
    if constexpr (NT) { const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p)); return make_float4(v[0], v[1], v[2], v[3]); }
    else return *reinterpret_cast<const float4*>(p);
}
template <bool NT> __device__ __forceinline__ void st4(float* p, const float4 v) {
    if constexpr (NT) { f32x4_nt o; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; __builtin_nontemporal_store(o, reinterpret_cast<f32x4_nt*>(p)); }
    else *reinterpret_cast<float4*>(p) = v;
}
template <typename T, bool GLP = false, int NTM = 0>
__global__ __launch_bounds__(256) void adamw_sn_kernel(const AdamDesc* adam, const SNDesc* sn, const WorkItem* items, float lr,
                                                      float b1, float b2, float eps, float wd, float bc1, float bc2sqrt,
                                                      double* gnorm_sq, const float* g_base = nullptr, const uint16_t* g_lp = nullptr, int desc_lp = 0) {
    constexpr int PITCH = 64 + (sizeof(T) == 2 ? 2 : 1);
    __shared__ T tile[64 * PITCH];
    __shared__ float tus[16][64];
    __shared__ float sm[4];
    const WorkItem it = items[blockIdx.x];
    const AdamDesc a = adam[it.desc];
    const SNDesc d = sn[a.sn];
    const int ct = (a.cols + 63) >> 6, rt = (a.rows + 63) >> 6;
    const int tap = it.chunk / (rt * ct);
    const int rem = it.chunk - tap * rt * ct;
    const int r0 = (rem / ct) << 6, c0 = (rem % ct) << 6;
    const float inv_sigma = d.sigma[1];
    float cdot = 0.f;
#pragma unroll
    for (int k = 0; k < SGV_DOT_SLOTS; ++k) cdot += d.dot[k];
    const int cq = threadIdx.x & 15, rr = threadIdx.x >> 4;
    const int col = c0 + cq * 4;
    const bool cok = col < a.cols;          // cols % 4 == 0 (checked on the host)
    float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) vv = *reinterpret_cast<const float4*>(d.v + (long)tap * a.cols + col);
    const long base = (long)tap * a.rows * a.cols;
    T* wc = reinterpret_cast<T*>(a.wc);
    const float step = lr / bc1, decay = 1.f - lr * wd;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, nacc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int rl = k * 16 + rr, row = r0 + rl;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cok && row < a.rows) {
            const long i = base + (long)row * a.cols + col;
            float4 g;
            if constexpr (GLP) {
                const uint2 raw = *reinterpret_cast<const uint2*>(g_lp + ((a.g + i) - g_base));       // 4 bf16, 8-byte aligned
                g.x = __builtin_bit_cast(float, raw.x << 16); g.y = __builtin_bit_cast(float, raw.x & 0xFFFF0000u);
                g.z = __builtin_bit_cast(float, raw.y << 16); g.w = __builtin_bit_cast(float, raw.y & 0xFFFF0000u);
            } else if (desc_lp && a.glp) {          // this layer's weight-gradient GEMM wrote bf16 (single-GPU path, engine option grad_bf16)
                const uint2 raw = *reinterpret_cast<const uint2*>(a.glp + i);
                g.x = __builtin_bit_cast(float, raw.x << 16); g.y = __builtin_bit_cast(float, raw.x & 0xFFFF0000u);
                g.z = __builtin_bit_cast(float, raw.y << 16); g.w = __builtin_bit_cast(float, raw.y & 0xFFFF0000u);
            } else {
                g = ld4<(NTM & 2) != 0>(a.g + i);
            }
            p = ld4<(NTM & 4) != 0>(a.p + i);
            float4 m = ld4<(NTM & 1) != 0>(a.m + i);
            float4 vs = ld4<(NTM & 1) != 0>(a.v + i);
            const float u_r = d.u[row];
            const float ur = u_r * cdot;
            g.x = (g.x - ur * vv.x) * inv_sigma; g.y = (g.y - ur * vv.y) * inv_sigma;
            g.z = (g.z - ur * vv.z) * inv_sigma; g.w = (g.w - ur * vv.w) * inv_sigma;
            nacc += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
#define SGV_ADAM1(F)                                                    \
            p.F *= decay;                                               \
            m.F = m.F * b1 + (1.f - b1) * g.F;                          \
            vs.F = vs.F * b2 + (1.f - b2) * g.F * g.F;                  \
            p.F -= step * (m.F / (sqrtf(vs.F) / bc2sqrt + eps));
            SGV_ADAM1(x) SGV_ADAM1(y) SGV_ADAM1(z) SGV_ADAM1(w)
#undef SGV_ADAM1
            st4<(NTM & 4) != 0>(a.p + i, p);
            st4<(NTM & 1) != 0>(a.m + i, m);
            st4<(NTM & 1) != 0>(a.v + i, vs);
            if constexpr (sizeof(T) == 2) {
                if (wc) {
                    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                    bf16x4_t o;
                    o[0] = (__bf16)p.x; o[1] = (__bf16)p.y; o[2] = (__bf16)p.z; o[3] = (__bf16)p.w;
                    if constexpr ((NTM & 8) != 0) {
                        typedef int i32x2_t __attribute__((ext_vector_type(2)));
                        __builtin_nontemporal_store(__builtin_bit_cast(i32x2_t, o), reinterpret_cast<i32x2_t*>(reinterpret_cast<__bf16*>(a.wc) + i));
                    } else *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(a.wc) + i) = o;
                }
            }
            t0 += p.x * u_r; t1 += p.y * u_r; t2 += p.z * u_r; t3 += p.w * u_r;
        }
        T* trow = tile + rl * PITCH + cq * 4;
        trow[0] = from_f32<T>(p.x); trow[1] = from_f32<T>(p.y); trow[2] = from_f32<T>(p.z); trow[3] = from_f32<T>(p.w);
    }
    tus[rr][cq * 4 + 0] = t0; tus[rr][cq * 4 + 1] = t1; tus[rr][cq * 4 + 2] = t2; tus[rr][cq * 4 + 3] = t3;
    const float w = wave_sum(nacc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) gnorm_sq[blockIdx.x] = (double)(sm[0] + sm[1] + sm[2] + sm[3]);      // per-item partial, summed in index order later
    if (threadIdx.x < 64 && c0 + (int)threadIdx.x < a.cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += tus[k][threadIdx.x];
        d.tpart[((long)(r0 >> 6) * a.taps + tap) * a.cols + c0 + threadIdx.x] = t;
    }
    if (a.wct) {
        T* dst = reinterpret_cast<T*>(a.wct) + (long)(a.taps - 1 - tap) * a.rows * a.cols;
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
        for (int r = ty; r < 64; r += 4) {
            const int cc = c0 + r, row = r0 + tx;
            if (row < a.rows && cc < a.cols) {
                if constexpr ((NTM & 8) != 0 && sizeof(T) == 2) __builtin_nontemporal_store(__builtin_bit_cast(unsigned short, tile[tx * PITCH + r]), reinterpret_cast<unsigned short*>(dst) + (long)cc * a.rows + row);
                else dst[(long)cc * a.rows + row] = tile[tx * PITCH + r];
            }
        }
    }
}
int opt_adamw_sn(const AdamDesc* adam_dev, const SNDesc* sn_dev, const WorkItem* items, int n, float lr, float b1, float b2,
                 float eps, float wd, float bc1, float bc2sqrt, double* gnorm_sq, int compute_dtype, hipStream_t s, const float* g_base, const void* g_lp, int desc_lp) {
    if (n <= 0) return 0;
    // default 9: the moments and the bf16 operand copies bypass the caches (measured, DESIGN.md section 13: moments 12.20 -> 12.05 ms
    // per step, the pass alone 2.90 -> 2.70 ms; the copies another 0.07 ms over eleven alternations)
    static const int ntm = getenv("SGV_ADAM_NT") ? atoi(getenv("SGV_ADAM_NT")) : 9;
    const uint16_t* lp = reinterpret_cast<const uint16_t*>(g_lp);
#define SGV_ADAM_LAUNCH(TT, GLP, MODE) hipLaunchKernelGGL((adamw_sn_kernel<TT, GLP, MODE>), dim3(n), dim3(256), 0, s, adam_dev, sn_dev, items, lr, b1, b2, eps, wd, bc1, bc2sqrt, gnorm_sq, g_base, lp, desc_lp)
#define SGV_ADAM_MODES(TT, GLP)                                   \
    switch (ntm) {                                                \
        case 0: SGV_ADAM_LAUNCH(TT, GLP, 0); break;               \
        case 3: SGV_ADAM_LAUNCH(TT, GLP, 3); break;               \
        case 5: SGV_ADAM_LAUNCH(TT, GLP, 5); break;               \
        case 1: SGV_ADAM_LAUNCH(TT, GLP, 1); break;               \
        default: SGV_ADAM_LAUNCH(TT, GLP, 9); break;              \
    }
    if (compute_dtype == 1) { if (lp) { SGV_ADAM_MODES(bf16_t, true) } else { SGV_ADAM_MODES(bf16_t, false) } }
    else { if (lp) { SGV_ADAM_MODES(float, true) } else { SGV_ADAM_MODES(float, false) } }
#undef SGV_ADAM_MODES
#undef SGV_ADAM_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int opt_adamw(const AdamDesc* adam_dev, const SNDesc* sn_dev, const WorkItem* items, int n, float lr, float b1, float b2,
              float eps, float wd, float bc1, float bc2sqrt, double* gnorm_sq, int compute_dtype, hipStream_t s, const float* gscale) {
    if (n > 0 && compute_dtype == 1) hipLaunchKernelGGL((adamw_kernel<true, true>), dim3(n), dim3(256), 0, s, adam_dev, sn_dev, items, lr, b1, b2, eps, wd, bc1, bc2sqrt, gnorm_sq, gscale);
    else if (n > 0) hipLaunchKernelGGL((adamw_kernel<true, false>), dim3(n), dim3(256), 0, s, adam_dev, sn_dev, items, lr, b1, b2, eps, wd, bc1, bc2sqrt, gnorm_sq, gscale);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int opt_grad_norm(const AdamDesc* adam_dev, const SNDesc* sn_dev, const WorkItem* items, int n, double* gnorm_sq,
                  hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL((adamw_kernel<false, false>), dim3(n), dim3(256), 0, s, adam_dev, sn_dev, items, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f, gnorm_sq, (const float*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- compute-dtype copies: wc[tap][r][c] = W[tap][r][c]; wct[tap][c][r] = W[taps-1-tap][r][c] ---------
// work item chunk = (tap, 32-row tile, 32-col tile)
template <typename T>
__global__ __launch_bounds__(256) void make_copies_kernel(const AdamDesc* adam, const WorkItem* items) {
    const WorkItem it = items[blockIdx.x];
    const AdamDesc a = adam[it.desc];
    const int ct = (a.cols + 31) >> 5, rt = (a.rows + 31) >> 5;
    const int tap = it.chunk / (rt * ct);
    const int rem = it.chunk - tap * rt * ct;
    const int r0 = (rem / ct) << 5, c0 = (rem % ct) << 5;
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* W = a.p + (long)tap * a.rows * a.cols;
    T* wc = reinterpret_cast<T*>(a.wc);
    T* wct = reinterpret_cast<T*>(a.wct);
    for (int r = ty; r < 32; r += 8) {
        const int rr = r0 + r, cc = c0 + tx;
        float x = 0.f;
        if (rr < a.rows && cc < a.cols) {
            x = W[(long)rr * a.cols + cc];
            if (wc) wc[(long)tap * a.rows * a.cols + (long)rr * a.cols + cc] = from_f32<T>(x);
        }
        tile[r][tx] = x;
    }
    if (!wct) return;
    __syncthreads();
    T* dst = wct + (long)(a.taps - 1 - tap) * a.rows * a.cols;
    for (int r = ty; r < 32; r += 8) {
        const int cc = c0 + r, rr = r0 + tx;
        if (rr < a.rows && cc < a.cols) dst[(long)cc * a.rows + rr] = from_f32<T>(tile[tx][r]);
    }
}
// transposed copy only, sourced from the compute copy AdamW just wrote (bf16) or the master (fp32 mode):
// wct[taps-1-tap][c][r] = w[tap][r][c]; 64x64 tiles
template <typename T>
__global__ __launch_bounds__(256) void make_wct_kernel(const AdamDesc* adam, const WorkItem* items) {
    const WorkItem it = items[blockIdx.x];
    const AdamDesc a = adam[it.desc];
    const int ct = (a.cols + 63) >> 6, rt = (a.rows + 63) >> 6;
    const int tap = it.chunk / (rt * ct);
    const int rem = it.chunk - tap * rt * ct;
    const int r0 = (rem / ct) << 6, c0 = (rem % ct) << 6;
    __shared__ T tile[64][64 + 8];
    const T* src = (a.wc ? reinterpret_cast<const T*>(a.wc) : reinterpret_cast<const T*>(a.p)) + (long)tap * a.rows * a.cols;
    T* dst = reinterpret_cast<T*>(a.wct) + (long)(a.taps - 1 - tap) * a.rows * a.cols;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int rr = r0 + r, cc = c0 + tx;
        tile[r][tx] = (rr < a.rows && cc < a.cols) ? src[(long)rr * a.cols + cc] : (T)0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int cc = c0 + r, rr = r0 + tx;
        if (rr < a.rows && cc < a.cols) dst[(long)cc * a.rows + rr] = tile[tx][r];
    }
}
int opt_make_wct(const AdamDesc* adam_dev, const WorkItem* items, int n, int compute_dtype, hipStream_t s) {
    if (n <= 0) return 0;
    if (compute_dtype == 1) hipLaunchKernelGGL((make_wct_kernel<bf16_t>), dim3(n), dim3(256), 0, s, adam_dev, items);
    else hipLaunchKernelGGL((make_wct_kernel<float>), dim3(n), dim3(256), 0, s, adam_dev, items);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int opt_make_copies(const AdamDesc* adam_dev, const WorkItem* items, int n, int compute_dtype, hipStream_t s) {
    if (n <= 0) return 0;
    if (compute_dtype == 1) hipLaunchKernelGGL((make_copies_kernel<bf16_t>), dim3(n), dim3(256), 0, s, adam_dev, items);
    else hipLaunchKernelGGL((make_copies_kernel<float>), dim3(n), dim3(256), 0, s, adam_dev, items);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
