// Plain (one-tap) NT GEMMs through hipBLASLt, for the shapes where the library beats the hand-written kernels
// (tests/micro/vendor_gemm.py: 3200x95008x1024 659 us vs 856 us; 3200x5120x1024 35 us vs 56 us).  Convolutions with
// taps > 1, the weight gradients, every fp32 GEMM and the K = 95 008 layers stay on gemm.hip (the library is slower there or
// has no such operation).  The library is resolved with dlopen (the copy torch already loaded when there is one), so
// libsgvae.so has no link-time dependency on it; when it is missing the callers keep using the hand-written kernels.
//
//   C[m][n] (row-major, ldc) = alpha * sum_k A[m][k] W[n][k] + bias[n]      alpha = 1/sigma (device) or 1
// is, in the library's column-major terms, D (N x M, ld = ldc) = alpha * op_T(W as K x N, ld = ldw) * (A as K x M, ld = lda)
// + bias along the rows of D: HIPBLASLT_EPILOGUE_BIAS with an fp32 bias.  The library takes a device-side alpha only as a
// VECTOR over the rows of D (HIPBLASLT_POINTER_MODE_ALPHA_DEVICE_VECTOR_BETA_HOST; the scalar device mode produced
// garbage on this version), so the caller passes p.scale_vec = N copies of 1/sigma (written by the power-iteration kernel
// that computes sigma); without a scale, alpha is the host constant 1.
#include "sgv_common.h"
#include <hipblaslt/hipblaslt.h>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <tuple>

namespace {
struct LtApi {
    void* h = nullptr;
    decltype(&hipblasLtCreate) Create = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
    decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
    decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
    decltype(&hipblasLtMatmul) Matmul = nullptr;
    hipblasLtHandle_t handle = nullptr;
    bool tried = false, ok = false;
};
LtApi g_lt;
std::mutex g_mu;

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr, ld = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool ok = false;
};
// one plan per (W, bias, alpha mode, shape, leading dimensions): the bias pointer is an attribute of the descriptor
typedef std::tuple<const void*, const void*, bool, int, int, int, long, long, long, long> PlanKey;
std::map<PlanKey, Plan> g_plans;
std::map<hipStream_t, void*> g_ws;       // one workspace per stream: launches on different streams may overlap
constexpr size_t kWorkspace = 64ull << 20;

bool lt_load() {
    if (g_lt.tried) return g_lt.ok;
    g_lt.tried = true;
    static const int on = getenv("SGV_VENDOR_GEMM") ? atoi(getenv("SGV_VENDOR_GEMM")) : 1;
    if (!on) return false;
    const char* names[] = {"libhipblaslt.so.1", "libhipblaslt.so", "/opt/rocm/lib/libhipblaslt.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return false;
#define SGV_LT_SYM(field, name) g_lt.field = (decltype(g_lt.field))dlsym(h, name); if (!g_lt.field) return false;
    SGV_LT_SYM(Create, "hipblasLtCreate")
    SGV_LT_SYM(LayoutCreate, "hipblasLtMatrixLayoutCreate")
    SGV_LT_SYM(DescCreate, "hipblasLtMatmulDescCreate")
    SGV_LT_SYM(DescSet, "hipblasLtMatmulDescSetAttribute")
    SGV_LT_SYM(PrefCreate, "hipblasLtMatmulPreferenceCreate")
    SGV_LT_SYM(PrefSet, "hipblasLtMatmulPreferenceSetAttribute")
    SGV_LT_SYM(PrefDestroy, "hipblasLtMatmulPreferenceDestroy")
    SGV_LT_SYM(DescDestroy, "hipblasLtMatmulDescDestroy")
    SGV_LT_SYM(LayoutDestroy, "hipblasLtMatrixLayoutDestroy")
    SGV_LT_SYM(Heuristic, "hipblasLtMatmulAlgoGetHeuristic")
    SGV_LT_SYM(Matmul, "hipblasLtMatmul")
#undef SGV_LT_SYM
    if (g_lt.Create(&g_lt.handle) != HIPBLAS_STATUS_SUCCESS) return false;
    g_lt.h = h;
    g_lt.ok = true;
    return true;
}

Plan* get_plan(const GemmNT& p) {
    const PlanKey key(p.W, p.bias, p.scale_vec != nullptr, p.M, p.N, p.K, p.lda, p.ldw, p.ldc, p.addend ? p.ldadd : 0L);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return &it->second;
    // plans are keyed by the weight / bias addresses: a caller that re-allocates its tensors (the operator API is handed torch
    // tensors) would otherwise grow the cache without bound -- past kMaxPlans the cache is dropped and rebuilt on demand
    constexpr size_t kMaxPlans = 512;
    if (g_plans.size() >= kMaxPlans) {
        for (auto& kv : g_plans) {
            Plan& q = kv.second;
            if (q.lc && q.lc != q.ld) g_lt.LayoutDestroy(q.lc);
            if (q.la) g_lt.LayoutDestroy(q.la);
            if (q.lb) g_lt.LayoutDestroy(q.lb);
            if (q.ld) g_lt.LayoutDestroy(q.ld);
            if (q.desc) g_lt.DescDestroy(q.desc);
        }
        g_plans.clear();
    }
    Plan pl;
    const int32_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
    const int32_t mode = p.scale_vec ? HIPBLASLT_POINTER_MODE_ALPHA_DEVICE_VECTOR_BETA_HOST : HIPBLASLT_POINTER_MODE_HOST;
    bool ok = g_lt.DescCreate(&pl.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && g_lt.DescSet(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && g_lt.DescSet(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && g_lt.DescSet(pl.desc, HIPBLASLT_MATMUL_DESC_POINTER_MODE, &mode, sizeof(mode)) == HIPBLAS_STATUS_SUCCESS;
    if (ok && p.bias) {
        const uint32_t epi = HIPBLASLT_EPILOGUE_BIAS;
        const int32_t btype = HIP_R_32F;
        const void* bp = p.bias;
        ok = g_lt.DescSet(pl.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS;
        ok = ok && g_lt.DescSet(pl.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &btype, sizeof(btype)) == HIPBLAS_STATUS_SUCCESS;
        ok = ok && g_lt.DescSet(pl.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) == HIPBLAS_STATUS_SUCCESS;
    }
    // "A" of the library = W stored [N][ldw] row-major = column-major K x N; "B" = activations, column-major K x M
    ok = ok && g_lt.LayoutCreate(&pl.la, HIP_R_16BF, (uint64_t)p.K, (uint64_t)p.N, p.ldw) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && g_lt.LayoutCreate(&pl.lb, HIP_R_16BF, (uint64_t)p.K, (uint64_t)p.M, p.lda) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && g_lt.LayoutCreate(&pl.ld, HIP_R_16BF, (uint64_t)p.N, (uint64_t)p.M, p.ldc) == HIPBLAS_STATUS_SUCCESS;
    if (p.addend) ok = ok && g_lt.LayoutCreate(&pl.lc, HIP_R_16BF, (uint64_t)p.N, (uint64_t)p.M, p.ldadd) == HIPBLAS_STATUS_SUCCESS;
    else pl.lc = pl.ld;
    if (ok) {
        hipblasLtMatmulPreference_t pref = nullptr;
        ok = g_lt.PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS;
        const uint64_t wsmax = kWorkspace;
        ok = ok && g_lt.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsmax, sizeof(wsmax)) == HIPBLAS_STATUS_SUCCESS;
        hipblasLtMatmulHeuristicResult_t res[1];
        int found = 0;
        ok = ok && g_lt.Heuristic(g_lt.handle, pl.desc, pl.la, pl.lb, pl.lc, pl.ld, pref, 1, res, &found) == HIPBLAS_STATUS_SUCCESS;
        ok = ok && found > 0 && res[0].state == HIPBLAS_STATUS_SUCCESS && res[0].workspaceSize <= kWorkspace;
        if (ok) { pl.algo = res[0].algo; pl.ws = res[0].workspaceSize; }
        if (pref) g_lt.PrefDestroy(pref);
    }
    pl.ok = ok;
    return &(g_plans[key] = pl);
}
}  // namespace

// Shapes handed to the library (bf16, one tap, bf16 output; a residual addend becomes the library's C with beta = 1): measured
// with tests/micro/vendor_vs_own.py, M = 3200: every N x K from 128 x 256 to 95008 x 1024 is 1.2-2x faster than the own
// kernel + split-K combine pass.
bool gemm_nt_vendor_eligible(int dtype, const GemmNT& p) {
    if (dtype != 1 || p.taps != 1 || p.out_f32 || p.gn_sums) return false;
    if (p.scale && !p.scale_vec) return false;                          // a device-side scale needs its vector form
    // narrower operands (the conditioner's 16-64 channel layers) measured slower through the library: 1074 vs 1091 samples/s
    if (p.K > 8192 || p.K < 128 || p.N < 128) return false;          // long contractions (1024 x 95008: 832 vs 1028 us): the LDS-DMA kernel wins
    return 2.0 * p.M * p.N * p.K >= 2.0e8;
}

// 0: launched; 1: library or plan unavailable (caller uses its own kernel); < 0: launch error
// frees the per-stream workspace of a stream that is about to be destroyed (sgv_destroy)
void gemm_nt_vendor_release_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ws.find(s);
    if (it == g_ws.end()) return;
    if (it->second) hipFree(it->second);
    g_ws.erase(it);
}

int launch_gemm_nt_vendor(const GemmNT& p, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!lt_load()) return 1;
    Plan* pl = get_plan(p);
    if (!pl->ok) return 1;
    void*& ws = g_ws[s];
    if (!ws && hipMalloc(&ws, kWorkspace) != hipSuccess) { ws = nullptr; return 1; }
    static const float one = 1.0f, zero = 0.0f;
    const float* alpha = p.scale_vec ? p.scale_vec : &one;
    const float* beta = p.addend ? &one : &zero;
    const hipblasStatus_t st = g_lt.Matmul(g_lt.handle, pl->desc, alpha, p.W, pl->la, p.A, pl->lb, beta, p.addend ? p.addend : p.C, pl->lc, p.C, pl->ld,
                                           &pl->algo, ws, kWorkspace, s);
    return st == HIPBLAS_STATUS_SUCCESS ? 0 : -2;
}
