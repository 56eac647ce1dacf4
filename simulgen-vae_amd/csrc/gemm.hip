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
#include "sgv_common.h"

// =========================================================================================
// NT
// =========================================================================================
template <typename T, int KCH>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmNT p) {
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
    const int tiles_n = (p.N + 127) >> 7;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm << 7, n0 = tn << 7;
    const int kchunks = (p.K + BK - 1) / BK;
    const int total = p.taps * kchunks;
    const int z = blockIdx.y;
    const int s_begin = (int)((long)total * z / p.splitk);
    const int s_end = (int)((long)total * (z + 1) / p.splitk);

    const int kq = tid % KCH;
    const int r0 = tid / KCH;
    const T* Ag = reinterpret_cast<const T*>(p.A);
    const T* Wg = reinterpret_cast<const T*>(p.W);
    const T* a_base[LPT];
    const T* w_base[LPT];
    int a_t[LPT];
    bool a_ok[LPT], w_ok[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int r = r0 + i * RSTEP;
        const int m = m0 + r, n = n0 + r;
        a_ok[i] = m < p.M;
        w_ok[i] = n < p.N;
        a_t[i] = m % p.Tlen;
        a_base[i] = Ag + (long)m * p.lda + kq * EPC;
        w_base[i] = Wg + (long)n * p.ldw + kq * EPC;
    }
    uint4 ra[LPT], rw[LPT];
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

    auto gload = [&](int s) {
        const int j = s / kchunks;
        const int kc = (s - j * kchunks) * BK;
        const int dt = j - p.pad;
        const bool kok = (kc + kq * EPC) < p.K;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const bool oka = a_ok[i] && kok && ((unsigned)(a_t[i] + dt) < (unsigned)p.Tlen);
            ra[i] = zero4;
            if (oka) ra[i] = *reinterpret_cast<const uint4*>(a_base[i] + (long)dt * p.lda + kc);
            const bool okw = w_ok[i] && kok;
            rw[i] = zero4;
            if (okw) rw[i] = *reinterpret_cast<const uint4*>(w_base[i] + (long)j * p.w_tap_stride + kc);
        }
    };
    auto sstore = [&](int buf) {
        unsigned char* sa = smem + buf * 2 * TILEB;
        unsigned char* sw = sa + TILEB;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int off = (r0 + i * RSTEP) * ROWB + kq * 16;
            if constexpr (IS_BF16) {
                *reinterpret_cast<uint4*>(sa + off) = ra[i];
                *reinterpret_cast<uint4*>(sw + off) = rw[i];
            } else {
                uint32_t* da = reinterpret_cast<uint32_t*>(sa + off);
                uint32_t* dw = reinterpret_cast<uint32_t*>(sw + off);
                da[0] = ra[i].x; da[1] = ra[i].y; da[2] = ra[i].z; da[3] = ra[i].w;
                dw[0] = rw[i].x; dw[1] = rw[i].y; dw[2] = rw[i].z; dw[3] = rw[i].w;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    auto compute = [&](int buf) {
        const unsigned char* sa = smem + buf * 2 * TILEB;
        const unsigned char* sw = sa + TILEB;
        if constexpr (IS_BF16) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = *reinterpret_cast<const bf16x8*>(sa + (wm * 64 + i * 32 + lr) * ROWB + (ks * 2 + lh) * 16);
                    bfr[i] = *reinterpret_cast<const bf16x8*>(sw + (wn * 64 + i * 32 + lr) * ROWB + (ks * 2 + lh) * 16);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                float af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = *reinterpret_cast<const float*>(sa + (wm * 64 + i * 32 + lr) * ROWB + (ks * 2 + lh) * 4);
                    bfr[i] = *reinterpret_cast<const float*>(sw + (wn * 64 + i * 32 + lr) * ROWB + (ks * 2 + lh) * 4);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    if (s_begin < s_end) {
        gload(s_begin);
        sstore(0);
        __syncthreads();
        int cur = 0;
        for (int s = s_begin; s < s_end; ++s) {
            const bool more = (s + 1) < s_end;
            if (more) gload(s + 1);
            compute(cur);
            if (more) sstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const float sc = p.scale ? *p.scale : 1.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn * 64 + b * 32 + lr;
            if (col >= p.N) continue;
            const float bv = (p.bias && p.splitk == 1) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= p.M) continue;
                float v = acc[a][b][r];
                if (p.splitk > 1) {
                    p.partial[((long)z * p.M + row) * p.N + col] = v;
                } else {
                    v = v * sc + bv;
                    if (p.addend) v += to_f32(reinterpret_cast<const T*>(p.addend)[(long)row * p.ldadd + col]);
                    if (p.out_f32) reinterpret_cast<float*>(p.C)[(long)row * p.ldc + col] = v;
                    else reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = from_f32<T>(v);
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
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTN p) {
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
    const int tiles_2 = (p.N2 + 127) >> 7;
    const int t1 = blockIdx.x / tiles_2, t2 = blockIdx.x - t1 * tiles_2;
    const int i0 = t1 << 7, j0 = t2 << 7;
    const int tap = blockIdx.y / p.splitk, z = blockIdx.y - tap * p.splitk;
    const int dt = tap - p.pad;
    const int ksteps = (p.M + KR - 1) / KR;
    const int s_begin = (int)((long)ksteps * z / p.splitk);
    const int s_end = (int)((long)ksteps * (z + 1) / p.splitk);

    const int cq = tid % CPR;
    const int r0 = tid / CPR;
    const T* Ag = reinterpret_cast<const T*>(p.A);
    const T* Bg = reinterpret_cast<const T*>(p.B);
    const bool a_cok = (i0 + cq * EPC) < p.N1;
    const bool b_cok = (j0 + cq * EPC) < p.N2;
    uint4 ra[LPT], rb[LPT];
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

    auto gload = [&](int s) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int m = s * KR + r0 + i * RSTEP;
            const bool mok = m < p.M;
            ra[i] = zero4;
            if (mok && a_cok) ra[i] = *reinterpret_cast<const uint4*>(Ag + (long)m * p.lda + i0 + cq * EPC);
            const int t = m % p.Tlen;
            rb[i] = zero4;
            if (mok && b_cok && ((unsigned)(t + dt) < (unsigned)p.Tlen))
                rb[i] = *reinterpret_cast<const uint4*>(Bg + (long)(m + dt) * p.ldb + j0 + cq * EPC);
        }
    };
    auto sstore = [&](int buf) {
        unsigned char* sa = smem + buf * 2 * TILEB;
        unsigned char* sb = sa + TILEB;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int off = (r0 + i * RSTEP) * ROWB + cq * 16;
            *reinterpret_cast<uint4*>(sa + off) = ra[i];
            *reinterpret_cast<uint4*>(sb + off) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    auto compute = [&](int buf) {
        const unsigned char* sa = smem + buf * 2 * TILEB;
        const unsigned char* sb = sa + TILEB;
        if constexpr (IS_BF16) {
#pragma unroll
            for (int ks = 0; ks < KR / 16; ++ks) {
                bf16x8 af[2], bfr[2];
                if constexpr (USE_TR) {
                    // ds_read_b64_tr_b16: per 16-lane group g a 4(k) x 16(col) block; lane 4q+p of the
                    // group addresses row q, cols 4p..4p+3; lane i receives column i, rows 0..3.
                    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
                    const int kb = ks * 16 + 8 * (g >> 1) + q;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int ca = wm * 64 + i * 32 + 16 * (g & 1) + 4 * pp;
                        const int cb = wn * 64 + i * 32 + 16 * (g & 1) + 4 * pp;
                        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                        const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sa + (kb * LD + ca) * 2));
                        const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sa + ((kb + 4) * LD + ca) * 2));
                        const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + (kb * LD + cb) * 2));
                        const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + ((kb + 4) * LD + cb) * 2));
                        s16x8 av, bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { av[e] = alo[e]; av[e + 4] = ahi[e]; bv[e] = blo[e]; bv[e + 4] = bhi[e]; }
                        af[i] = __builtin_bit_cast(bf16x8, av);
                        bfr[i] = __builtin_bit_cast(bf16x8, bv);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int k = ks * 16 + 8 * lh + e;
                            af[i][e] = *reinterpret_cast<const bf16_t*>(sa + (k * LD + wm * 64 + i * 32 + lr) * 2);
                            bfr[i][e] = *reinterpret_cast<const bf16_t*>(sb + (k * LD + wn * 64 + i * 32 + lr) * 2);
                        }
                    }
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KR / 2; ++ks) {
                float af[2], bfr[2];
                const int k = ks * 2 + lh;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = *reinterpret_cast<const float*>(sa + (k * LD + wm * 64 + i * 32 + lr) * 4);
                    bfr[i] = *reinterpret_cast<const float*>(sb + (k * LD + wn * 64 + i * 32 + lr) * 4);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    if (s_begin < s_end) {
        gload(s_begin);
        sstore(0);
        __syncthreads();
        int cur = 0;
        for (int s = s_begin; s < s_end; ++s) {
            const bool more = (s + 1) < s_end;
            if (more) gload(s + 1);
            compute(cur);
            if (more) sstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

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

int gemm_nt_pick_splitk(int M, int N, int K, int taps, int dtype) {
    const int bk = dtype == 1 ? 32 : 16;
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    const int total = taps * cdiv(K, bk);
    if (tiles >= 384) return 1;
    int want = cdiv(768, tiles);
    int cap = total / 8;
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want > 32) want = 32;
    return want < 1 ? 1 : want;
}

int gemm_tn_pick_splitk(int M, int N1, int N2, int taps, int dtype) {
    const int kr = dtype == 1 ? 32 : 16;
    const int tiles = cdiv(N1, 128) * cdiv(N2, 128) * taps;
    const int total = cdiv(M, kr);
    if (tiles >= 384) return 1;
    int want = cdiv(768, tiles);
    int cap = total / 4;
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want > 32) want = 32;
    return want < 1 ? 1 : want;
}

int launch_gemm_nt(int dtype, const GemmNT& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return 0;
    const int epc = dtype == 1 ? 8 : 4;
    if (p.K % epc || p.lda % epc || p.ldw % epc || p.w_tap_stride % epc) return -1;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return -1;
    if (p.splitk > 1 && !p.partial) return -1;
    dim3 grid(cdiv(p.M, 128) * cdiv(p.N, 128), p.splitk);
    if (dtype == 1) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, 4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<float, 4>), grid, dim3(256), 0, s, p);
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
    dim3 grid(cdiv(p.N1, 128) * cdiv(p.N2, 128), p.taps * p.splitk);
    if (dtype == 1) {
        if (p.use_tr) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, false>), grid, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((gemm_tn_kernel<float, false>), grid, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
