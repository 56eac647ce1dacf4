// Parameter-set object of the operator-level ABI (include/sgvae_ops.h): the multi-tensor passes of optim.hip (legacy
// spectral-norm power iteration, <G,W> dots, gradient 2-norm, AdamW with the spectral-norm chain rule) over an arbitrary
// list of caller-owned fp32 tensors, so a host-side model (the latent conditioner mirror) spends 8 launches per step on
// its parameters instead of ~5 per tensor.  torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW semantics
// (reference modules/latent_conditioner.py:195-211,304,314).
#include "../../include/sgvae_ops.h"
#include "sgv_ew.h"

#include <math.h>
#include <string.h>
#include <vector>

int sgv_set_error(int code, const char* fmt, ...);   // engine.hip

struct sgv_pset {
    std::vector<SNDesc> sn;
    std::vector<AdamDesc> adam;
    std::vector<int> sn_of_entry;                 // entry -> index into sn, or -1
    SNDesc* sn_dev = nullptr; AdamDesc* adam_dev = nullptr;
    WorkItem *items_sn = nullptr, *items_dot = nullptr, *items_adam = nullptr, *items_ts = nullptr, *items_ss = nullptr;
    int n_items_sn = 0, n_items_dot = 0, n_items_adam = 0, n_items_ts = 0, n_items_ss = 0;
    float *mv = nullptr, *sigma = nullptr, *dots = nullptr, *tmp = nullptr, *coef = nullptr, *dot_part = nullptr;
    double *gnorm = nullptr, *gnorm_part = nullptr;      // gnorm_part: one partial per AdamW work item, summed in index order (no atomics)
    std::vector<FinDot> fin_dots;
    size_t n_tmp = 0, n_dots = 0;
    int step = 0;
};

__global__ void pset_clip_coef_kernel(const double* sumsq, float max_norm, float* out) {
    const float tn = (float)sqrt(sumsq[0]);
    out[0] = max_norm > 0.f ? fminf(1.f, max_norm / (tn + 1e-6f)) : 1.f;
    out[1] = tn;
}

static size_t al4(size_t n) { return (n + 3) / 4 * 4; }

extern "C" {

int sgv_pset_destroy(sgv_pset* ps) {
    if (!ps) return 0;
    void* ptrs[] = {ps->sn_dev, ps->adam_dev, ps->items_sn, ps->items_dot, ps->items_adam, ps->items_ts, ps->items_ss, ps->mv, ps->sigma, ps->dots, ps->tmp, ps->coef, ps->gnorm, ps->dot_part, ps->gnorm_part};
    for (void* p : ptrs) if (p) hipFree(p);
    delete ps;
    return 0;
}

int sgv_pset_create(const sgv_pset_entry* entries, int n, sgv_pset** out) {
    if (!entries || n <= 0 || !out) return sgv_set_error(-1, "sgv_pset_create: bad argument");
    sgv_pset* ps = new sgv_pset();
    size_t total = 0, n_sn = 0, n_tmp = 0;
    for (int i = 0; i < n; ++i) {
        const sgv_pset_entry& e = entries[i];
        if (!e.p || !e.g || e.n <= 0 || e.n % 4 || ((uintptr_t)e.p & 15) || ((uintptr_t)e.g & 15)) {
            delete ps;
            return sgv_set_error(-1, "sgv_pset_create: entry %d needs 16-byte aligned p/g and a multiple of 4 elements (n=%ld)", i, e.n);
        }
        if (e.rows > 0) {
            if ((long)e.rows * e.cols != e.n || e.cols % 4 || !e.u || !e.v) {
                delete ps;
                return sgv_set_error(-1, "sgv_pset_create: spectral-norm entry %d needs rows*cols == n, cols %% 4 == 0 and u, v", i);
            }
            ++n_sn;
            n_tmp += al4(e.cols) + al4(e.rows);
            n_tmp += al4((size_t)((e.rows + SN_ROWS_PER_ITEM - 1) / SN_ROWS_PER_ITEM) * e.cols) + al4((size_t)((e.cols + SN_COLS_PER_ITEM - 1) / SN_COLS_PER_ITEM) * e.rows);
        }
        total += al4(e.n);
    }
#define PS_ALLOC(ptr, bytes)                                                                   \
    do {                                                                                       \
        if (hipMalloc((void**)&(ptr), (bytes) ? (bytes) : 256) != hipSuccess || hipMemset((ptr), 0, (bytes) ? (bytes) : 256) != hipSuccess) { \
            sgv_pset_destroy(ps);                                                              \
            return sgv_set_error(-2, "sgv_pset_create: allocation of %zu bytes failed", (size_t)(bytes)); \
        }                                                                                      \
    } while (0)
    PS_ALLOC(ps->mv, total * 2 * sizeof(float));
    PS_ALLOC(ps->sigma, (n_sn ? n_sn : 1) * 2 * sizeof(float));
    ps->n_dots = (n_sn ? n_sn : 1) * SGV_DOT_SLOTS;
    PS_ALLOC(ps->dots, ps->n_dots * sizeof(float));
    ps->n_tmp = n_tmp;
    PS_ALLOC(ps->tmp, n_tmp * sizeof(float));
    PS_ALLOC(ps->coef, 2 * sizeof(float));
    PS_ALLOC(ps->gnorm, sizeof(double));
    std::vector<WorkItem> i_sn, i_dot, i_adam, i_ts, i_ss;
    size_t off = 0, toff = 0;
    ps->sn_of_entry.assign(n, -1);
    for (int i = 0; i < n; ++i) {
        const sgv_pset_entry& e = entries[i];
        AdamDesc a; memset(&a, 0, sizeof(a));
        a.p = e.p; a.g = e.g; a.m = ps->mv + off; a.v = ps->mv + total + off; a.n = e.n; a.sn = -1; a.rows = 1; a.cols = (int)e.n; a.taps = 1;
        off += al4(e.n);
        if (e.rows > 0) {
            SNDesc d; memset(&d, 0, sizeof(d));
            const int si = (int)ps->sn.size();
            d.W = e.p; d.u = e.u; d.v = e.v;
            d.tmp_t = ps->tmp + toff; toff += al4(e.cols);
            d.tmp_s = ps->tmp + toff; toff += al4(e.rows);
            d.tpart = ps->tmp + toff; toff += al4((size_t)((e.rows + SN_ROWS_PER_ITEM - 1) / SN_ROWS_PER_ITEM) * e.cols);
            d.spart = ps->tmp + toff; toff += al4((size_t)((e.cols + SN_COLS_PER_ITEM - 1) / SN_COLS_PER_ITEM) * e.rows);
            d.sigma = ps->sigma + 2 * si; d.dot = ps->dots + (size_t)si * SGV_DOT_SLOTS; d.G = e.g;
            d.taps = 1; d.rows = e.rows; d.cols = e.cols; d.active = 1;
            ps->sn.push_back(d);
            ps->sn_of_entry[i] = si;
            a.sn = si; a.rows = e.rows; a.cols = e.cols;
            const int rb = (e.rows + SN_ROWS_PER_ITEM - 1) / SN_ROWS_PER_ITEM, cb = (e.cols + SN_COLS_PER_ITEM - 1) / SN_COLS_PER_ITEM;
            for (int c = 0; c < rb * cb; ++c) i_sn.push_back({si, c});
            for (int c = 0; c < (e.cols + 63) / 64; ++c) i_ts.push_back({si, c});
            for (int c = 0; c < (e.rows + 63) / 64; ++c) i_ss.push_back({si, c});
            ps->fin_dots.push_back({(const float*)(uintptr_t)i_dot.size(), d.dot, (int)((e.n + OPT_CHUNK - 1) / OPT_CHUNK), 0});
            for (long c = 0; c < (e.n + OPT_CHUNK - 1) / OPT_CHUNK; ++c) i_dot.push_back({si, (int)c});
        }
        const int id = (int)ps->adam.size();
        ps->adam.push_back(a);
        for (long c = 0; c < (e.n + OPT_CHUNK - 1) / OPT_CHUNK; ++c) i_adam.push_back({id, (int)c});
    }
    auto up = [&](const void* src, size_t bytes, void** dst) -> bool {
        if (bytes == 0) { *dst = nullptr; return true; }
        return hipMalloc(dst, bytes) == hipSuccess && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(ps->sn.data(), sizeof(SNDesc) * ps->sn.size(), (void**)&ps->sn_dev) || !up(ps->adam.data(), sizeof(AdamDesc) * ps->adam.size(), (void**)&ps->adam_dev) ||
        !up(i_sn.data(), sizeof(WorkItem) * i_sn.size(), (void**)&ps->items_sn) || !up(i_dot.data(), sizeof(WorkItem) * i_dot.size(), (void**)&ps->items_dot) ||
        !up(i_adam.data(), sizeof(WorkItem) * i_adam.size(), (void**)&ps->items_adam) ||
        !up(i_ts.data(), sizeof(WorkItem) * i_ts.size(), (void**)&ps->items_ts) || !up(i_ss.data(), sizeof(WorkItem) * i_ss.size(), (void**)&ps->items_ss)) {
        sgv_pset_destroy(ps);
        return sgv_set_error(-2, "sgv_pset_create: table upload failed");
    }
    ps->n_items_sn = (int)i_sn.size(); ps->n_items_dot = (int)i_dot.size(); ps->n_items_adam = (int)i_adam.size();
    ps->n_items_ts = (int)i_ts.size(); ps->n_items_ss = (int)i_ss.size();
    PS_ALLOC(ps->dot_part, (i_dot.size() ? i_dot.size() : 1) * sizeof(float));
    PS_ALLOC(ps->gnorm_part, (i_adam.size() ? i_adam.size() : 1) * sizeof(double));
    for (auto& f : ps->fin_dots) f.src = ps->dot_part + (size_t)(uintptr_t)f.src;
    *out = ps;
    return 0;
}

int sgv_pset_power_iteration(sgv_pset* ps, int train, void* stream) {
    if (!ps) return sgv_set_error(-1, "null parameter set");
    if (ps->sn.empty()) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (opt_sn_power_iteration(ps->sn_dev, ps->items_sn, ps->n_items_sn, ps->items_sn, ps->n_items_sn, ps->items_ts, ps->n_items_ts,
                               ps->items_ss, ps->n_items_ss, (int)ps->sn.size(), train, s))
        return sgv_set_error(-2, "power-iteration launch failed");
    return 0;
}

const float* sgv_pset_sigma(const sgv_pset* ps, int entry) {
    if (!ps || entry < 0 || entry >= (int)ps->sn_of_entry.size() || ps->sn_of_entry[entry] < 0) return nullptr;
    return ps->sigma + 2 * ps->sn_of_entry[entry];
}

int sgv_pset_step(sgv_pset* ps, float lr, float weight_decay, float max_norm, float* total_norm_host, void* stream) {
    if (!ps) return sgv_set_error(-1, "null parameter set");
    hipStream_t s = (hipStream_t)stream;
    ps->step += 1;
    const double b1 = 0.9, b2 = 0.999;
    const float bc1 = (float)(1.0 - pow(b1, (double)ps->step)), bc2s = (float)sqrt(1.0 - pow(b2, (double)ps->step));
    // <G,W>/sigma per spectrally-normalised tensor and the gradient norm: per-work-item partials, summed in a fixed order
    if (opt_sn_grad_dot(ps->sn_dev, ps->items_dot, ps->n_items_dot, ps->dot_part, s)) return sgv_set_error(-2, "grad-dot launch failed");
    if (!ps->fin_dots.empty()) ew_fin_dots(ps->fin_dots.data(), (int)ps->fin_dots.size(), s);
    if (opt_grad_norm(ps->adam_dev, ps->sn_dev, ps->items_adam, ps->n_items_adam, ps->gnorm_part, s)) return sgv_set_error(-2, "grad-norm launch failed");
    ew_rowsum_d(ps->gnorm_part, ps->n_items_adam, 1, ps->gnorm, 1.0, s);
    hipLaunchKernelGGL(pset_clip_coef_kernel, dim3(1), dim3(1), 0, s, ps->gnorm, max_norm, ps->coef);
    if (opt_adamw(ps->adam_dev, ps->sn_dev, ps->items_adam, ps->n_items_adam, lr, (float)b1, (float)b2, 1e-8f, weight_decay, bc1, bc2s, ps->gnorm_part, 0, s, ps->coef))
        return sgv_set_error(-2, "adamw launch failed");
    if (total_norm_host) {
        float h[2] = {0.f, 0.f};
        if (hipMemcpyAsync(h, ps->coef, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return sgv_set_error(-2, "norm read-back failed");
        *total_norm_host = h[1];
    }
    return 0;
}

}  // extern "C"
