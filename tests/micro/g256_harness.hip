// Standalone correctness + timing harness for the GEMM kernels (no Python, no torch: seconds of GPU-box time).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DT256_PINGPONG=0 ...] tests/micro/g256_harness.hip -o tests/micro/_ab/g256
//   ./g256 [check=1] [reps=5] M N K taps splitk [M N K taps splitk ...]      (splitk 0: the launcher's own choice)
// Per shape: the 256x256 persistent kernel (gemm256.hip) and the round-1 kernels (gemm.hip: 128x256 wide64p / 128x128) on the
// same random bf16 operands, each checked against a plain one-thread-per-output fp32 reference kernel; with stats=1 the
// fused GroupNorm statistics are checked against sums over the stored output.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../simulgen-vae_amd/csrc/gemm.hip"
#include "../../simulgen-vae_amd/csrc/gemm256.hip"

bool gemm_nt_vendor_eligible(int, const GemmNT&) { return false; }
int launch_gemm_nt_vendor(const GemmNT&, hipStream_t) { return 1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill_bf16(bf16_t* p, long n, uint32_t seed, float amp) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (bf16_t)(((float)(x & 0xFFFFFF) / 8388608.0f - 1.0f) * amp);
    }
}
__global__ void fill_f32(float* p, long n, uint32_t seed, float amp) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((float)(x & 0xFFFFFF) / 8388608.0f - 1.0f) * amp;
    }
}
// reference: one thread per output, fp32 accumulation in k order
__global__ void ref_kernel(const bf16_t* A, const bf16_t* W, const float* bias, float sc, const bf16_t* addend, float* out, int M, int N, int K,
                           int taps, int pad, int Tlen) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx - (long)m * N);
    const int t = m % Tlen;
    float acc = 0.f;
    for (int j = 0; j < taps; ++j) {
        const int tt = t + j - pad;
        if (tt < 0 || tt >= Tlen || m + j - pad >= M) continue;      // rows past M read as zero (partial last sample)
        const bf16_t* a = A + (long)(m + j - pad) * K;
        const bf16_t* w = W + ((long)j * N + n) * K;
        for (int k = 0; k < K; k += 8) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + k);
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (float)av[e] * (float)wv[e];
        }
    }
    float v = acc * sc + (bias ? bias[n] : 0.f);
    out[idx] = v;
}
__global__ void cmp_kernel(const bf16_t* C, const float* ref, const bf16_t* addend, long n, float* maxerr, float* maxref) {
    float me = 0.f, mr = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float r = (float)(bf16_t)ref[i];
        if (addend) r = (float)(bf16_t)(r + (float)addend[i]);
        me = fmaxf(me, fabsf((float)C[i] - r));
        mr = fmaxf(mr, fabsf(r));
    }
    for (int o = 32; o > 0; o >>= 1) { me = fmaxf(me, __shfl_xor(me, o, 64)); mr = fmaxf(mr, __shfl_xor(mr, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMax((int*)maxerr, __float_as_int(me)); atomicMax((int*)maxref, __float_as_int(mr)); }
}

static float time_it(int reps, hipStream_t s, const std::function<int()>& f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) if (f()) { printf("launch rejected\n"); return -1.f; }
    CK(hipStreamSynchronize(s));
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a, s)); f(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    int check = 1, reps = 5, stats = 0, addend_on = 0, old_on = 1, Tlen = 200, autop = 0;
    std::vector<int> nums;
    for (int i = 1; i < argc; ++i) {
        if (!strncmp(argv[i], "check=", 6)) check = atoi(argv[i] + 6);
        else if (!strncmp(argv[i], "reps=", 5)) reps = atoi(argv[i] + 5);
        else if (!strncmp(argv[i], "stats=", 6)) stats = atoi(argv[i] + 6);
        else if (!strncmp(argv[i], "addend=", 7)) addend_on = atoi(argv[i] + 7);
        else if (!strncmp(argv[i], "old=", 4)) old_on = atoi(argv[i] + 4);
        else if (!strncmp(argv[i], "T=", 2)) Tlen = atoi(argv[i] + 2);
        else if (!strncmp(argv[i], "auto=", 5)) autop = atoi(argv[i] + 5);
        else nums.push_back(atoi(argv[i]));
    }
    if (nums.size() % 5) { printf("usage: g256 [check=1] [reps=5] [stats=0] [addend=0] [old=1] [T=200] M N K taps splitk ...\n"); return 1; }
    hipStream_t s; CK(hipStreamCreate(&s));
    int bad = 0;
    for (size_t c = 0; c < nums.size(); c += 5) {
        const int M = nums[c], N = nums[c + 1], K = nums[c + 2], taps = nums[c + 3];
        int sk = nums[c + 4];
        const int pad = (taps - 1) / 2;
        bf16_t *A, *W, *C, *C2, *AD = nullptr; float *bias, *scale, *ref = nullptr, *partial = nullptr, *gpart = nullptr, *me;
        double* gsums = nullptr;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)taps * N * K * 2));
        CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&C2, (size_t)M * N * 2));
        CK(hipMalloc(&bias, (size_t)N * 4)); CK(hipMalloc(&scale, 4)); CK(hipMalloc(&me, 8));
        fill_bf16<<<2048, 256, 0, s>>>(A, (long)M * K, 0x1234u, 1.0f);
        fill_bf16<<<2048, 256, 0, s>>>(W, (long)taps * N * K, 0x9876u, 0.05f);
        fill_f32<<<64, 256, 0, s>>>(bias, N, 0x55u, 0.5f);
        const float sc = 0.75f;
        CK(hipMemcpyAsync(scale, &sc, 4, hipMemcpyHostToDevice, s));
        if (addend_on) { CK(hipMalloc(&AD, (size_t)M * N * 2)); fill_bf16<<<2048, 256, 0, s>>>(AD, (long)M * N, 0x777u, 1.0f); }
        CK(hipMemsetAsync(C, 0xFF, (size_t)M * N * 2, s)); CK(hipMemsetAsync(C2, 0xFF, (size_t)M * N * 2, s));
        GemmNT p; memset(&p, 0, sizeof(p));
        p.A = A; p.lda = K; p.W = W; p.ldw = K; p.w_tap_stride = (long)N * K; p.C = C; p.ldc = N; p.bias = bias; p.scale = scale;
        p.addend = AD; p.ldadd = N;
        p.M = M; p.N = N; p.K = K; p.taps = taps; p.pad = pad; p.Tlen = Tlen;
        const int G = 8;
        GemmPlan pl = {1, 1, 1, M, 0};
        const size_t cap = (size_t)80 << 20;       // floats of split-K workspace, as the engine has
        if (autop) {
            CK(hipMalloc(&partial, cap * 4)); p.partial = partial;
            pl = gemm_nt_plan(1, p, cap, 0);
            sk = pl.sk_main;
        } else {
            if (sk <= 0) sk = gemm_nt256_pick_splitk(M, N, K, taps);
            if (sk > 1) { CK(hipMalloc(&partial, (size_t)sk * M * N * 4)); p.partial = partial; }
        }
        p.splitk = sk;
        const bool do_stats = stats && sk == 1 && N % G == 0 && (N / G) % 4 == 0 && N / G >= 64 && Tlen >= 128;
        if (do_stats) {
            CK(hipMalloc(&gpart, gemm_nt256_part_floats(M, N, 1) * 4));
            const int B = (M + Tlen - 1) / Tlen;
            CK(hipMalloc(&gsums, (size_t)B * G * 2 * 8));
            p.gn_part = gpart; p.gn_sums = gsums; p.gn_Cg = N / G; p.gn_G = G;
        }
        const double flop = 2.0 * M * N * K * taps;
        fprintf(stderr, "[%d %d %d %d] t256...\n", M, N, K, taps);
        const float t_new = time_it(reps, s, [&]() { return autop ? launch_gemm_nt_planned(1, p, pl, s) : launch_gemm_nt256(p, s); });
        // round-1 kernels
        GemmNT po = p; po.C = C2; po.gn_part = nullptr; po.gn_sums = nullptr;
        po.splitk = gemm_nt_pick_splitk(M, N, K, taps, 1);
        float* partial_o = nullptr;
        float t_old = -1.f;
        fprintf(stderr, "r1...\n");
        if (old_on) {
            if (po.splitk > 1) { CK(hipMalloc(&partial_o, (size_t)po.splitk * M * N * 4)); po.partial = partial_o; }
            t_old = time_it(reps, s, [&]() { return launch_gemm_nt(1, po, s); });
        }
        float err_new = -1.f, err_old = -1.f, mref = 0.f, serr = -1.f;
        fprintf(stderr, "check...\n");
        if (check) {
            CK(hipMalloc(&ref, (size_t)M * N * 4));
            const long tot = (long)M * N;
            ref_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(A, W, bias, sc, AD, ref, M, N, K, taps, pad, Tlen);
            float hm[2];
            CK(hipMemsetAsync(me, 0, 8, s));
            cmp_kernel<<<1024, 256, 0, s>>>(C, ref, AD, tot, me, me + 1);
            CK(hipMemcpyAsync(hm, me, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            err_new = hm[0]; mref = hm[1];
            if (old_on) {
                CK(hipMemsetAsync(me, 0, 8, s));
                cmp_kernel<<<1024, 256, 0, s>>>(C2, ref, AD, tot, me, me + 1);
                CK(hipMemcpyAsync(hm, me, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
                err_old = hm[0];
            }
            if (do_stats) {
                const int B = (M + Tlen - 1) / Tlen, Cg = N / G;
                std::vector<uint16_t> hc((size_t)M * N);
                std::vector<double> hs((size_t)B * G * 2);
                CK(hipMemcpy(hc.data(), C, (size_t)M * N * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hs.data(), gsums, hs.size() * 8, hipMemcpyDeviceToHost));
                std::vector<double> rs((size_t)B * G * 2, 0.0);
                for (int m = 0; m < M; ++m)
                    for (int n = 0; n < N; ++n) {
                        uint32_t u = (uint32_t)hc[(size_t)m * N + n] << 16; float f; memcpy(&f, &u, 4);
                        double* d = &rs[((size_t)(m / Tlen) * G + n / Cg) * 2];
                        d[0] += f; d[1] += (double)f * f;
                    }
                double worst = 0.0;
                for (size_t i = 0; i < rs.size(); i += 2) {
                    const double cnt = (double)Cg * Tlen;
                    worst = std::max(worst, fabs(hs[i] - rs[i]) / (sqrt(rs[i + 1] * cnt) + 1e-30));     // |dsum| / (rms * count)
                    worst = std::max(worst, fabs(hs[i + 1] - rs[i + 1]) / (rs[i + 1] + 1e-30));
                }
                serr = (float)worst;
            }
        }
        const bool ok_new = !check || (err_new >= 0.f && err_new <= 0.012f * mref + 1e-3f);
        const bool ok_old = !check || !old_on || (err_old <= 0.012f * mref + 1e-3f);
        const bool ok_st = serr < 0.f || serr < 2e-5f;
        if (!ok_new || !ok_st) bad = 1;
        if (autop) printf("[plan kind=%d sk=%d/%d m_main=%d] ", pl.kind, pl.sk_main, pl.sk_tail, pl.m_main);
        printf("M=%d N=%d K=%d taps=%d | t256 sk=%d %8.1f us %7.1f TF/s err %.3g/%.3g %s%s", M, N, K, taps, sk, t_new * 1e3, flop / t_new / 1e9,
               err_new, mref, ok_new ? "OK" : "FAIL", serr >= 0.f ? (ok_st ? " stats OK" : " stats FAIL") : "");
        if (serr >= 0.f) printf("(%.2g)", serr);
        if (old_on) printf(" | r1 sk=%d %8.1f us %7.1f TF/s err %.3g %s", po.splitk, t_old * 1e3, flop / t_old / 1e9, err_old, ok_old ? "OK" : "FAIL");
        printf("\n"); fflush(stdout);
        hipFree(A); hipFree(W); hipFree(C); hipFree(C2); hipFree(bias); hipFree(scale); hipFree(me);
        if (AD) hipFree(AD); if (ref) hipFree(ref); if (partial) hipFree(partial); if (partial_o) hipFree(partial_o);
        if (gpart) hipFree(gpart); if (gsums) hipFree(gsums);
    }
    return bad;
}
