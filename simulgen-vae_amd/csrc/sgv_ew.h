// Host-callable launchers of the elementwise / normalisation / optimizer kernels (ew.hip, optim.hip).
#pragma once
#include "sgv_common.h"

struct GNParams {
    const void* y = nullptr;    long ldy = 0;      // conv output (pre-norm) [B*T][C]
    const void* res = nullptr;  long ldres = 0;    // optional residual base
    void* out = nullptr;        long ldout = 0;    // result
    const void* dout = nullptr; long lddout = 0;   // incoming gradient, or loss target (FROM_LOSS)
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float* dgamma = nullptr;
    float* dbeta = nullptr;
    float* dbias = nullptr;                        // bias gradient of the producing conv
    float* part = nullptr;                         // workspace for per-block column sums (ew_gn_part_floats)
    float* cdot = nullptr;                         // <G, W_eff> = sum dY*(y - cbias) of the producing conv (slot 0 of its dot slots)
    float* cdot_part = nullptr;                    // deferred mode: one partial per block of the dY-producing launch goes here and the
    int* cdot_blocks = nullptr;                    //   caller sums them later (ew_fin_dots); *cdot_blocks (host) = how many were written.
                                                   //   null: the launcher sums them into cdot[0] itself (workspace: p.part)
    float* ptot = nullptr;                         // deferred mode: [B][3][C] per-sample column totals (sum dz, sum dz*xhat, bias-gradient
                                                   //   term) for ew_fin_affine; null: the launcher finalises dbeta / dgamma / dbias itself
    const float* cbias = nullptr;                  // that conv's bias
    const float* yf32 = nullptr; long ldyf = 0;    // act mode 2: the conv output (fp32) paired with dY in `y`
    double* sums = nullptr;                        // [B*G][2] sum, sum of squares
    double* sums2 = nullptr;                       // [B*G][2] backward group sums
    double* loss_sums = nullptr;                   // [2]
    int accum_affine = 0;                          // immediate mode: dbeta / dgamma / dbias += instead of =
    int defer_colsum = 0;                          // ew_act modes 1 / 2: leave the per-block column sums in `part` ([ew_act_part_rows][C]); the caller sums them later (ew_fin_affine, arrays = 1)
    float* lpart = nullptr;                        // recon loss: per-block (selected, squared) partial sums (workspace)
    int B = 0, T = 0, C = 0, G = 1, Cg = 1, CV = 1;
    float rscale = 1.f;                            // residual / incoming-gradient scale
    float gscale = 1.f;                            // output gradient scale (loss weight)
    int loss_type = 0;
};

// second operand of ew_gn_tail: out = relu(A + gn(y)), A = gn2(y2) (gamma2 / beta2 / sums2) or y2 * cscale[b][c] (cscale != null)
struct GNTail {
    const void* y2 = nullptr; long ldy2 = 0;
    const float* gamma2 = nullptr;
    const float* beta2 = nullptr;
    double* sums2 = nullptr;            // [B*G][2] statistics of y2 (same groups as y)
    const float* cscale = nullptr;      // [B][C]
};

constexpr int SGV_GN_MAX_GROUPS = 32;
// convgn.hip: fused small convolution + GroupNorm + GELU (+ residual) forward, one workgroup per (group, sample)
struct ConvGN {
    const void* A; long lda;            // input [B*T][lda] bf16, K columns used
    const void* W; long ldw;            // weights [taps][N][ldw] bf16 (K contiguous)
    long w_tap_stride;
    const float* bias;                  // [N]
    const float* scale;                 // device scalar 1/sigma, or null
    void* y; long ldy;                  // pre-norm conv output [B*T][ldy] bf16 (stored)
    void* out; long ldout;              // result
    const void* res; long ldres;        // optional residual base
    float rscale;
    const float* gamma; const float* beta;
    double* sums;                       // [B*G][2] sum, sum of squares (stored)
    int B, T, N, K, taps, pad, G, Cg;
};
// backward mirror: input gradient of the UPPER convolution + GroupNorm / GELU backward of the stage below it in one launch
// (the gradient wrt the lower stage's activated output is never stored)
struct ConvGNBwd {
    const void* A; long lda;            // dY of the upper stage [B*T][lda] bf16, K = its output channels
    const void* W; long ldw;            // its transposed, tap-flipped weight copy [taps][N][ldw] bf16
    long w_tap_stride;
    const float* scale;                 // 1/sigma of the upper convolution, or null
    const void* addend; long ldadd;     // optional [B*T][ldadd] bf16 added to the input gradient before it is rounded (residual path)
    const void* premul; long ldpre;     // optional [B*T][ldpre] bf16 x: the (rounded) input gradient is multiplied by gelu'(x) and rounded
                                        //   again (the upper convolution reads GELU(x): modules/decoder.py condition blocks)
    const void* y; long ldy;            // pre-norm output of the lower stage [B*T][ldy] bf16
    const double* sums;                 // its forward statistics [B*G][2]
    const float* gamma; const float* beta;
    const float* cbias;                 // bias of the lower convolution (for <G, W_eff>), or null
    void* dy; long lddy;                // out: gradient wrt y [B*T][lddy] bf16
    void* da; long ldda;                // optional out: the input gradient itself [B*T][ldda] bf16 (a residual block below needs it again)
    double* sums2;                      // out [B*G][2]
    float* ptot;                        // out [B][3][C]: per-sample column totals (sum dz, sum dz*xhat, bias-gradient term)
    float* cdot_part;                   // out [B*G]: partials of <G, W_eff> = sum dY * (y - cbias), or null
    float rscale, gscale;
    int B, T, N, K, taps, pad, G, Cg;
};
bool conv_gn_bwd_eligible(int dtype, const ConvGNBwd& p);
int launch_conv_gn_bwd(const ConvGNBwd& p, hipStream_t s);
bool conv_gn_fused_eligible(int dtype, const ConvGN& p);
int launch_conv_gn_fwd(const ConvGN& p, hipStream_t s);
// Deterministic reductions: no kernel of the step accumulates floating-point values with atomics.  Block partials go to
// workspaces and are summed in a fixed order, either by the launcher itself or, for quantities nobody needs before the
// optimizer (GroupNorm affine / bias gradients, <G, W_eff>), by two table-driven passes the engine runs once per bucket.
struct FinDot { const float* src; float* dst; int count; int pad; };                      // dst[0] = sum src[0..count)
struct FinAffine { const float* ptot; float* dbeta; float* dgamma; float* dbias; int C; int B; int accum; int arrays; };   // d*[c] (+)= sum_b ptot[b][k][c]; arrays = k range (0 -> 3; 1: ptot is [B][C], dbeta only)
int ew_fin_dots(const FinDot* items_host, int n, hipStream_t s);
int ew_fin_affine(const FinAffine* items_host, int n, hipStream_t s);
// out[b*n + j] = scale * sum_{r<R} part[(b*R + r)*n + j] in a fixed order (out_f or out_d, the other null)
int ew_rowsum(const float* part, int batches, int R, int n, float* out_f, double* out_d, double scale, hipStream_t s);
int ew_rowsum_d(const double* part, int R, int n, double* out_d, double scale, hipStream_t s);
int ew_gn_stats(int dtype, GNParams p, hipStream_t s);
// whole GroupNorm passes: one fused launch when a (sample, group) slab is small, else the multi-kernel path
int ew_gn_fwd(int dtype, int act, GNParams p, hipStream_t s);   // stats (p.sums zeroed by the caller) + apply
int ew_gn_bwd(int dtype, int act, GNParams p, hipStream_t s);   // reduce + finalize + dY; act in {0, 1 gelu, 3 relu}
int ew_gn_bwd_reduce_act(int dtype, int act, GNParams p, hipStream_t s);   // act: 0 none, 1 gelu, 3 relu
int ew_gn_bwd_apply_act(int dtype, int act, GNParams p, hipStream_t s);
int ew_gn_apply(int dtype, int act, GNParams p, hipStream_t s);
int ew_gn_tail(int dtype, GNParams p, GNTail t, hipStream_t s);    // p: y / gamma / beta / sums (given) / out; see GNTail
size_t ew_gn_part_floats(int B, int T, int C);
int ew_gn_max_blocks(int B, int T, int C);
int ew_act_part_rows(int B, int T, int C);         // rows of per-block column sums an ew_act launch with dbias writes          // upper bound of the per-block <G,W_eff> partials one launch writes
int ew_recon_loss(int dtype, int train, GNParams p, hipStream_t s);
int ew_recon_bwd_apply(int dtype, GNParams p, hipStream_t s);
int ew_act(int dtype, int mode, GNParams p, hipStream_t s);
int ew_add3(int dtype, const void* a, long lda, const void* b, long ldb, const void* c, long ldc, void* out, long ldo,
            int rows, int C, hipStream_t s);
int ew_transpose(int src_dtype, int dst_dtype, const void* src, void* dst, int Bn, int I, int J, long lds_, long ldd,
                 long sbatch, long dbatch, hipStream_t s);
int ew_randn(float* out, long n, uint64_t seed, uint64_t stream, hipStream_t s, long per_row = 0, int world = 1, int rank = 0);
int ew_latent_fwd(const float* last, const float* eps, float* z, int B, int Z, double* kl_sum, hipStream_t s);
int ew_latent_bwd(const float* last, const float* eps, const float* dz, float* dlast, int B, int Z, float coef, hipStream_t s);
int ew_stage_fwd(int dtype, const float* pz, const float* qz, const float* eps, const void* dec_out, long ldd,
                 void* zs_next, long ldz, float* zmap, int M, int C, float std_scale, double* kl_sum, float inv_b,
                 double* kl_part, hipStream_t s);      // kl_part: workspace of >= 2048 doubles (per-block partials)
int ew_stage_bwd(int dtype, const float* pz, const float* qz, const float* eps, const void* dzs, long ldd, void* g_p,
                 void* g_q, int M, int C, float coef, hipStream_t s);
int ew_linear_head_fwd(int xdtype, const void* X, const float* W, const float* bias, const float* scale, float* Y, int B,
                       int K, int O, float* part, hipStream_t s);   // part: workspace of >= 128 * B * O floats (K-slice partials)
int ew_linear_head_bwd(int xdtype, const float* dY, const void* X, const float* W, const float* scale, const void* addend,
                       void* dX, float* dW, float* db, int B, int K, int O, hipStream_t s);
int ew_linear_expand_fwd(int dtype, const float* X, const float* W, const float* bias, const float* scale, void* Y, int B,
                         int K, int O, hipStream_t s);
int ew_linear_expand_bwd(int dtype, const void* dY, const float* X, const float* W, const float* scale, float* dX, float* dW,
                         float* db, int B, int K, int O, hipStream_t s);
int ew_augment(int dtype, const void* data, void* out, long sample_elems, int batch, const int* idx,
               const unsigned long long* noise_seed, const float* scale, const int* mix_idx, const float* lam, hipStream_t s);
int ew_pack_bf16(const float* src, void* dst_bf16, long n, hipStream_t s);     // data-parallel gradient payload: fp32 -> bf16 (RNE)
int ew_unpack_bf16(const void* src_bf16, float* dst, long n, hipStream_t s);
int ew_axpy(float* y, const float* x, float a, long n, hipStream_t s);
int ew_scale3(float* d0, float* d1, float* d2, const float* x, float a, long n, hipStream_t s);   // d_k = a * x[k*n .. (k+1)*n)
int ew_fill_from_scalar(float* dst, const float* src_scalar, long n, hipStream_t s);   // dst[i] = *src_scalar
int ew_scale(float* y, float a, long n, hipStream_t s);
// input pipeline
constexpr int SGV_MINMAX_ROWSPLIT = 64;
int ew_minmax_fit(const float* rows, long n_rows, int N, float* mn, float* mx, float* partial, int RS, int accumulate, hipStream_t s);
int ew_minmax_coeffs(const float* mn, const float* mx, int N, float lo, float hi, float* scale, float* offset, hipStream_t s);
int ew_scale_convert(int dtype, const float* src, const float* scale, const float* offset, void* dst, long n_rows, int N, hipStream_t s);
int ew_cast_rows(int dtype, const float* src, long lds_, void* dst, long ldd, int rows, int C, hipStream_t s);

// ---------------- optimizer / spectral norm (optim.hip) ----------------
// One descriptor per spectrally-normalised weight, internal layout [taps][rows][cols] fp32.
struct SNDesc {
    float* W;          // master weight
    float* u;          // [rows]
    float* v;          // [taps*cols], internal order (tap, col)
    float* tmp_t;      // [taps*cols] scratch: W^T u
    float* tmp_s;      // [rows]      scratch: W v
    float* tpart;      // [ceil(rows/64)][taps*cols] partials of W^T u per 64-row block (summed in block order by sn_tsum_kernel)
    float* spart;      // [taps*ceil(cols/1024)][rows] partials of W v per (tap, 1024-column block) (summed by sn_ssum_kernel)
    float* sigma;      // [2]: sigma, 1/sigma
    float* dot;        // [SGV_DOT_SLOTS] partial <G, W_eff> (lives in the gradient arena's small zone: all-reduced with it)
    const float* G;    // gradient wrt W_eff (null if the layer gets no gradient)
    const void* wc;    // bf16 copy of W in the same [taps][rows][cols] order (bf16 engines), or null: W v reads it instead of W
    int taps, rows, cols;
    int active;        // participates in this forward
};
// One descriptor per trainable tensor for the fused AdamW pass.
struct AdamDesc {
    float* p; float* g; float* m; float* v;
    long n;
    int sn;            // index into the SNDesc table, or -1 (bias / GroupNorm affine)
    int rows, cols;    // SN geometry for the rank-1 correction (taps*rows*cols == n)
    void* wc; void* wct;  // compute-dtype copies [taps][rows][cols] and [taps'][cols][rows] (tap-flipped), or null
    int taps;
    const unsigned short* glp;   // bf16 mirror of g (same element order) or null: read instead of g when the launch says so (desc_lp)
};
struct WorkItem { int desc; int chunk; };
// <G,W_eff> of a layer lives in this many slots of the gradient arena's small zone; the AdamW pass adds them up.  The
// fixed-order finalize (ew_fin_dots) writes the whole value to slot 0, the other slots stay at their initial zero -- the slot count is the arena
// layout round 1's atomic accumulation chose, kept so that checkpoints and the all-reduced arena keep their offsets.
constexpr int SGV_DOT_SLOTS = 32;

// items_ts / items_ss: (desc, 64-element chunk of taps*cols / 1024-row chunk) for the fixed-order partial sums
int opt_sn_power_iteration(const SNDesc* descs_dev, const WorkItem* items1, int n1, const WorkItem* items3, int n3,
                           const WorkItem* items_ts, int n_ts, const WorkItem* items_ss, int n_ss, int ndesc, int train, hipStream_t s);
// dot_part[i] = <G, W> / sigma of work item i (summed per layer by ew_fin_dots)
int opt_sn_grad_dot(const SNDesc* descs_dev, const WorkItem* items, int n, float* dot_part, hipStream_t s);
int opt_adamw(const AdamDesc* adam_dev, const SNDesc* sn_dev, const WorkItem* items, int n, float lr, float b1, float b2,
              float eps, float wd, float bc1, float bc2sqrt, double* gnorm_part, int compute_dtype, hipStream_t s, const float* gscale = nullptr);
// gnorm_part: one double per work item (the block's sum of squared gradients), written at gnorm_part[blockIdx.x]: pass the
// table base + the offset of `items` in the table; the caller sums the whole table in index order (ew_rowsum_d)
// 64x64-tile AdamW for spectrally-normalised conv weights; also writes wc/wct and accumulates W_new^T u into tmp_t
int opt_adamw_sn(const AdamDesc* adam_dev, const SNDesc* sn_dev, const WorkItem* items, int n, float lr, float b1, float b2,
                 float eps, float wd, float bc1, float bc2sqrt, double* gnorm_sq, int compute_dtype, hipStream_t s,
                 const float* g_base = nullptr, const void* g_lp = nullptr, int desc_lp = 0);      // g_lp: gradients from the bf16 wire copy (offsets relative to g_base)
int opt_grad_norm(const AdamDesc* adam_dev, const SNDesc* sn_dev, const WorkItem* items, int n, double* gnorm_sq,
                  hipStream_t s);
int opt_make_copies(const AdamDesc* adam_dev, const WorkItem* items, int n, int compute_dtype, hipStream_t s);
int opt_make_wct(const AdamDesc* adam_dev, const WorkItem* items, int n, int compute_dtype, hipStream_t s);
constexpr int OPT_CHUNK = 8192;      // elements per work item in the flat passes
constexpr int SN_ROWS_PER_ITEM = 64; // rows per work item in the GEMV passes
constexpr int SN_COLS_PER_ITEM = 1024;
