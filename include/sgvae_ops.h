/* libsgvae.so -- operator-level C ABI for the image latent conditioner.
 *
 * The reference model (modules/latent_conditioner_model_cnn.py:28-362) is a ResNet-style CNN followed by two small
 * MLP heads; its training loop is modules/latent_conditioner.py:213-386.  The MI355X build keeps the layer graph and
 * the hand-derived backward on the host (simulgen-vae_amd/modules/latent_conditioner_model_cnn.py) and runs every
 * tensor operation through the stateless entry points below.  Conventions: all pointers are device pointers owned by
 * the caller; feature maps are channels-last [B][H][W][C] in the compute dtype (SGV_DTYPE_F32 = 0, SGV_DTYPE_BF16 = 1),
 * P = H*W; tensors of shape [B][features] and every parameter / gradient are fp32; work is enqueued on `stream`
 * (a hipStream_t, NULL = default stream) with no host synchronisation; return 0 on success, else a negative
 * SGV_ERR_* code with the text in sgv_last_error().  There is no CPU fallback. */
#ifndef SGVAE_OPS_H
#define SGVAE_OPS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* nn.Conv2d (latent_conditioner_model_cnn.py:93,97,103,186): lowered to im2col + GEMM.
 * col[(b,oh,ow)][(kh*KW+kw)*C + c], rows zero-padded to Kp = roundup(KH*KW*C, 8); weights are expected as
 * [Cout][KH][KW][Cin] rows padded the same way.  forward: im2col, gemm_nt; weight gradient: gemm_tn(dY, col);
 * input gradient: gemm_nt(dY, W^T) then col2im (gather, deterministic).  A 1x1 stride-1 convolution needs no im2col. */
int sgv_op_conv_out_shape(int H, int W, int C, int KH, int KW, int stride, int pad, int* Ho, int* Wo, int* Kp);
int sgv_op_im2col(int dtype, const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int stride, int pad, void* stream);
int sgv_op_col2im(int dtype, const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad, void* stream);
/* C[M][N] = scale[0] * A[M][K] . W[N][K]^T (+ bias[N]) (+ addend[M][N]) on the MFMA kernels of the VAE path
 * (K, N multiples of 8; scale/bias/addend may be NULL; out_f32 = 1 writes fp32 instead of the compute dtype). */
int sgv_op_gemm_nt(int dtype, const void* A, const void* W, void* C, const float* bias, const float* scale, const void* addend,
                   int M, int N, int K, int out_f32, void* stream);
/* The stem (nn.Conv2d(1, C, 7, padding=3), model_cnn.py:92-93) without an im2col matrix, bf16 only: x [B][H][W] one channel,
 * wp the packed weights [N][roundup(KH*KW, 8)], y [B][H][W][N] = scale[0] * conv(x, w) (stride 1, square odd window <= 7x7,
 * pad = (KH-1)/2), and the GroupNorm statistics of y (as stored) in sums (B*G*2 doubles: per-block partials in `part`,
 * sgv_op_stem_conv_workspace_floats() floats, combined in a fixed order). */
size_t sgv_op_stem_conv_workspace_floats(int B, int H, int W, int N);
int sgv_op_stem_conv_fwd(const void* x, const void* wp, const float* scale, void* y, double* sums, float* part, int B, int H, int W, int N,
                         int KH, int KW, int pad, int G, void* stream);
/* Its weight gradient, again without the im2col matrix: dW[n][kh*KW + kw] (fp32, rows padded with zeros to roundup(KH*KW, 8),
 * the layout sgv_op_gemm_tn(dy, col) gives) = sum_{b,h,w} dy[b][h][w][n] * x[b][h - pad + kh][w - pad + kw]; part: the same
 * workspace size as the forward. */
int sgv_op_stem_conv_dw(const void* x, const void* dy, float* dW, float* part, int B, int H, int W, int N, int KH, int KW, int pad, void* stream);
/* The same convolution as an implicit GEMM (no im2col matrix; bf16 or fp32, Cin and N multiples of 8, KH*KW <= 31):
 *   y[b][oh][ow][n] = scale[0] * sum_{kh,kw,c} x[b][oh*stride - pad + kh][ow*stride - pad + kw][c] * Wt(kh*KW + kw)[n][c]
 * with x [B][H][W][Cin] and y [B][Ho][Wo][N] channels-last, Ho = (H + 2 pad - KH)/stride + 1 (pixels outside the image are
 * zero), and tap t's [N][Cin] weight matrix at W + t*w_tap_stride with row pitch ldw (elements; flip = 1: at
 * W + (KH*KW-1-t)*w_tap_stride).  Forward on the packed [Cout][KH][KW][Cin] weights: ldw = KH*KW*Cin, w_tap_stride = Cin.
 * Input gradient of a stride-1 convolution (model_cnn.py conv2 of the stride-1 blocks): x = dY, N = Cin, the transposed
 * weights [(kh,kw,ci)][Cout] with ldw = Cout, w_tap_stride = Cin*Cout, flip = 1, pad = KH-1-pad. */
int sgv_op_conv2d_nt(int dtype, const void* x, const void* W, void* y, const float* scale, int B, int H, int Wd, int Cin, int N,
                     int KH, int KW, int stride, int pad, long ldw, long w_tap_stride, int flip, void* stream);
/* C[M][N] = scale[0] * A[M][K] . W[N][K]^T + up2(addend) (bf16): the rows of C are the pixels of [B][H][W] images and addend is a
 * half-resolution batch [B][ceil(H/2)][ceil(W/2)][N] added at the even pixels only -- ResidualBlock's block-input gradient when
 * the skip projection has stride 2: conv1's input-gradient GEMM takes the (compact) input gradient of the 1x1 stride-2
 * projection as its addend instead of a zero-filled full-resolution copy (sgv_op_col2im).  N < 256 or K < 2048. */
int sgv_op_gemm_nt_add_s2(int dtype, const void* A, const void* W, void* C, const float* scale, const void* addend, int M, int N, int K,
                          int H, int Wd, void* stream);
/* dW[N1][N2] (fp32) = A[M][N1]^T . B[M][N2]  (N1, N2 multiples of 8).  The reduction runs over M = B*H*W rows:
 * sgv_op_gemm_tn_splitk() returns the number of row slices to use, the caller passes that many N1*N2 fp32 slabs
 * (slabs may be NULL when splitk == 1). */
int sgv_op_gemm_tn_splitk(int dtype, int M, int N1, int N2);
/* Weight gradient of the convolution without the im2col matrix (the X operand of the TN GEMM is addressed through the
 * window geometry): dW[n1][(kh*KW + kw)*Cin + c] (fp32) = sum_{b,oh,ow} dy[b][oh][ow][n1] * x[b][oh*stride - pad + kh][ow*stride - pad + kw][c];
 * dy [B][Ho][Wo][N1], x [B][H][W][Cin] channels-last, Cin and N1 multiples of 8; splitk / slabs as for sgv_op_gemm_tn with
 * M = B*Ho*Wo, N2 = KH*KW*Cin. */
int sgv_op_conv2d_tn(int dtype, const void* dy, const void* x, float* dW, int B, int H, int Wd, int Cin, int N1, int KH, int KW,
                     int stride, int pad, float* slabs, int splitk, void* stream);
int sgv_op_gemm_tn(int dtype, const void* A, const void* Bm, float* dW, int M, int N1, int N2, float* slabs, int splitk, void* stream);

/* nn.GroupNorm + optional ReLU (model_cnn.py:94,98,104,187-188; act: 0 none, 3 relu) on [B][P][C], C % 8 == 0,
 * G <= 32.  sums: B*G*2 doubles written by the forward and read by the backward; sums2: same size scratch;
 * part: sgv_op_gn_workspace_floats() floats scratch (forward and backward: per-block partial sums, combined in a fixed order --
 * no atomics, results are bitwise reproducible); dgamma/dbeta are ACCUMULATED into (+=). */
int sgv_op_gn_fwd(int dtype, int act, const void* y, void* out, int B, int P, int C, int G, const float* gamma, const float* beta,
                  double* sums, float* part, void* stream);
/* out = act(gn(y)) with the statistics given (sums as written by sgv_op_gn_fwd / sgv_op_stem_conv_fwd). */
int sgv_op_gn_apply(int dtype, int act, const void* y, void* out, int B, int P, int C, int G, const float* gamma, const float* beta,
                    double* sums, void* stream);
/* Tail of ResidualBlock.forward (model_cnn.py:108-124) in one streaming pass after the statistics:
 *   out = relu(A + gn(y; gamma, beta))   with  A = gn(y2; gamma2, beta2)   (cscale == NULL: blocks without squeeze-excite)
 *                                          or  A = y2 * cscale[b][c]       (y2 = the normalised main branch, cscale [B][C])
 * y = skip-projection output, y2 as above, all [B][P][C]; sums (and sums2 when cscale == NULL) are WRITTEN (B*G*2 doubles each,
 * read later by sgv_op_gn_bwd).  Every term is rounded to the compute dtype before the sum, i.e. the result equals
 * sgv_op_gn_fwd (x2) / sgv_op_chan_scale_fwd followed by sgv_op_add_relu_fwd bit for bit. */
int sgv_op_gn_tail(int dtype, const void* y, const float* gamma, const float* beta, double* sums, const void* y2, const float* gamma2,
                   const float* beta2, double* sums2, const float* cscale, void* out, int B, int P, int C, int G, float* part, void* stream);
size_t sgv_op_gn_workspace_floats(int B, int P, int C);
int sgv_op_gn_bwd(int dtype, int act, const void* y, const void* dout, void* dy, int B, int P, int C, int G, const float* gamma,
                  const float* beta, double* sums, double* sums2, float* part, float* dgamma, float* dbeta, void* stream);
/* sgv_op_gn_bwd with dgamma / dbeta WRITTEN (=) instead of accumulated (a layer applied once per step needs no zero-fill). */
int sgv_op_gn_bwd_set(int dtype, int act, const void* y, const void* dout, void* dy, int B, int P, int C, int G, const float* gamma,
                      const float* beta, double* sums, double* sums2, float* part, float* dgamma, float* dbeta, void* stream);

/* nn.MaxPool2d(3, 2, 1) (model_cnn.py:189); argmax: one byte per output element (window position of the first
 * maximum, may be NULL in the forward when no backward follows). */
int sgv_op_maxpool_fwd(int dtype, const void* x, void* y, unsigned char* argmax, int B, int H, int W, int C, void* stream);
/* The stem's GroupNorm + ReLU + MaxPool (model_cnn.py:94-96,189) in one pass over the convolution output y [B][H][W][C]:
 * out = maxpool(relu(gn(y; sums, gamma, beta))) with every normalised value rounded to the compute dtype first (the result and
 * argmax equal sgv_op_gn_apply followed by sgv_op_maxpool_fwd); sums as left by sgv_op_gn_fwd / sgv_op_stem_conv_fwd,
 * coef: 2*B*C floats of scratch.  The normalised activation itself is not stored: the backward (sgv_op_maxpool_bwd,
 * sgv_op_gn_bwd with act = relu) needs only argmax, y and sums. */
int sgv_op_gn_relu_maxpool_fwd(int dtype, const void* y, const double* sums, const float* gamma, const float* beta, int G, void* out,
                               unsigned char* argmax, float* coef, int B, int H, int W, int C, void* stream);
int sgv_op_maxpool_bwd(int dtype, const unsigned char* argmax, const void* dy, void* dx, int B, int H, int W, int C, void* stream);
/* out = relu(a + b) (model_cnn.py:131-132) and d = dout * (out > 0); plain add for gradient joins. */
int sgv_op_add_relu_fwd(int dtype, const void* a, const void* b, void* out, long n, void* stream);
int sgv_op_relu_bwd(int dtype, const void* out, const void* dout, void* d, long n, void* stream);
int sgv_op_add(int dtype, const void* a, const void* b, void* out, long n, void* stream);
/* AdaptiveAvgPool2d(1) (model_cnn.py:40,48,214): y[b][c] = mean_p x; backward dx (+)= dy / P. */
int sgv_op_avgpool_fwd(int dtype, const void* x, float* y, int B, int P, int C, void* stream);
int sgv_op_avgpool_bwd(int dtype, const float* dy, void* dx, int B, int P, int C, int accumulate, void* stream);
/* SqueezeExcitation scaling x * y.view(b,c,1,1) (model_cnn.py:51): out = x * s[b][c]; backward dx = dout*s,
 * ds[b][c] = sum_p dout*x. */
int sgv_op_chan_scale_fwd(int dtype, const void* x, const float* s, void* out, int B, int P, int C, void* stream);
int sgv_op_chan_scale_bwd(int dtype, const void* x, const float* s, const void* dout, void* dx, float* ds, int B, int P, int C, void* stream);

/* Small fp32 layers of the SE blocks and heads (model_cnn.py:41-50,220-277).
 * linear: y = act(scale[0] * x W^T + bias), act: 0 none, 1 relu, 2 sigmoid; act_bwd: dz = dy * act'(y) from the stored
 * output; linear_bwd: dx (= or +=) scale * dz W, dW = scale * dz^T x, db = sum_b dz (dx, db may be NULL; dW and db NULL: dx only). */
int sgv_op_linear_fwd(const float* x, const float* W, const float* bias, const float* scale, float* y, int B, int K, int O, int act, void* stream);
int sgv_op_act_fwd(const float* x, float* y, long n, int act, void* stream);
int sgv_op_act_bwd(const float* y, const float* dy, float* dz, long n, int act, void* stream);
int sgv_op_linear_bwd(const float* dz, const float* x, const float* W, const float* scale, float* dx, int accumulate_dx, float* dW, float* db,
                      int B, int K, int O, void* stream);
/* nn.LayerNorm(K) (eps 1e-5); stat: 2*B floats (mean, rstd) written by the forward. */
int sgv_op_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stat, int B, int K, void* stream);
int sgv_op_layernorm_bwd(const float* x, const float* gamma, const float* stat, const float* dy, float* dx, float* dgamma, float* dbeta,
                         int B, int K, void* stream);
/* nn.BatchNorm1d(K) (eps 1e-5, momentum 0.1): train = batch statistics + running-buffer update, eval = running
 * statistics; stat: 2*K floats (mean, rstd used). */
int sgv_op_batchnorm_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float* y, float* stat,
                         int B, int K, int train, void* stream);
int sgv_op_batchnorm_bwd(const float* x, const float* gamma, const float* stat, const float* dy, float* dx, float* dgamma, float* dbeta,
                         int B, int K, int train, void* stream);
/* out = a * mask * scale (nn.Dropout with an injected 0/1 mask and scale 1/(1-p); mask NULL = all ones), fp32 add,
 * and nn.MSELoss: loss_dev[0] = mean((pred-target)^2) (double), dpred = gscale * dloss/dpred (may be NULL). */
int sgv_op_mask_scale(const float* a, const float* mask, float scale, float* out, long n, void* stream);
int sgv_op_addf(const float* a, const float* b, float* out, long n, void* stream);
int sgv_op_mse(const float* pred, const float* target, double* loss_dev, float* dpred, float gscale, long n, void* stream);
/* End-to-end conditioner loop (latent_conditioner_e2e.py:66-92,244-256,374): the value (mean reduction, double, no
 * gradient -- the reference detaches this term through numpy) of kind 0 nn.MSELoss, 1 nn.L1Loss, 2 nn.HuberLoss(delta),
 * 3 nn.SmoothL1Loss(beta = delta) between two fp32 arrays; and sklearn MinMaxScaler.inverse_transform on a
 * [rows][cols] fp32 array, y = (x - min_[c]) / scale_[c]. */
/* Host-model plumbing: table_dev = n_rows x {const float* src, float* dst, int64 count} on the device; copies every row
 * (one launch for all parameter gradients of a step). */
int sgv_op_multi_copy(const void* table_dev, int n_rows, void* stream);
int sgv_op_loss_value(int kind, const float* a, const float* b, double* loss_dev, float delta, long n, void* stream);
int sgv_op_cols_sub_div(const float* x, const float* col_min, const float* col_scale, float* y, long rows, int cols, void* stream);
/* Input augmentation of the training loop (latent_conditioner.py:107-159,261-279) on [B][H][W] fp32 images, random
 * draws made by the caller: flip_roll = torch.flip(dims=[2]) where flip[b] then torch.roll by (shift_x[b], shift_y[b]);
 * affine_sample = F.affine_grid(theta [B][2][3], align_corners=False) + F.grid_sample(bilinear, 'border', False)
 * (small rotation / scaling); mixup_rows: out[b] = lam*x[b] + (1-lam)*x[perm[b]] on rows of n floats. */
int sgv_op_flip_roll(const float* x, float* out, int B, int H, int W, const int* flip, const int* shift_x, const int* shift_y, void* stream);
int sgv_op_affine_sample(const float* x, float* out, int B, int H, int W, const float* theta, void* stream);
int sgv_op_mixup_rows(const float* x, const int* perm, float lam, float* out, int B, long n, void* stream);

/* Parameter side.  Legacy spectral norm (modules/common.py:15-37 -> torch nn/utils/spectral_norm.py): with the weight
 * as a [rows][cols] matrix, v = l2_normalize(W^T u), u = l2_normalize(W v) (eps 1e-12), sigma = u.(W v); the
 * mat-vecs are sgv_op_linear_fwd (W v) and sgv_op_matvec_t (W^T u).  sgv_op_dot writes {a.b, 1/(a.b)}.
 * sgv_op_sn_grad: gradient wrt weight_orig from the gradient G wrt W/sigma: (G - (<G,W_orig>/sigma) u v^T) / sigma,
 * gw = {<G,W_orig>}, sigma2 = {sigma, 1/sigma}.  conv_weight_pack/unpack: reference [Cout][Cin][KH][KW] fp32 <-> the
 * GEMM layout [Cout][(kh*KW+kw)*Cin+ci] padded to a multiple of 8 (compute dtype / fp32).
 * Gradient clipping (latent_conditioner.py:304): sumsq accumulates sum g^2 into a double, clip_coef writes
 * {min(1, max_norm/(norm+1e-6)), norm}; sgv_op_adamw = torch.optim.AdamW on one tensor with g scaled by gscale[0]. */
int sgv_op_l2_normalize(const float* x, float* out, long n, float eps, void* stream);
int sgv_op_dot(const float* a, const float* b, float* out4, long n, void* stream);     /* out4: {a.b, 1/(a.b), 8 bytes of scratch} */
int sgv_op_matvec_t(const float* W, const float* x, float* out, int rows, int cols, void* stream);   /* out = W^T x */
int sgv_op_sn_grad(const float* G, const float* u, const float* v, const float* gw, const float* sigma2, float* out, int rows, int cols,
                   void* stream);
int sgv_op_conv_weight_pack(int dtype, const float* w, void* packed, int Cout, int Cin, int KH, int KW, void* stream);
int sgv_op_conv_weight_unpack(const float* packed, float* w, int Cout, int Cin, int KH, int KW, void* stream);
int sgv_op_sumsq(const float* g, long n, double* acc, void* stream);
int sgv_op_clip_coef(const double* sumsq, float max_norm, float* out2, void* stream);
int sgv_op_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                 int step, const float* gscale, void* stream);
/* Parameter set: the same steps for a whole list of tensors in a handful of launches (multi-tensor kernels of the VAE
 * path).  Entry: parameter p and its gradient buffer g (fp32, n elements, n % 4 == 0, 16-byte aligned, both owned by
 * the caller and at fixed addresses); rows > 0 marks a spectrally normalised [rows][cols] weight (cols % 4 == 0) with
 * its u [rows] / v [cols] vectors -- for those, g must hold the gradient wrt W/sigma and the step applies the chain
 * rule.  Adam moments live inside the object.
 *   sgv_pset_power_iteration: one legacy power iteration for every normalised weight (train != 0 updates u, v), sigma
 *     = u.(W v); sgv_pset_sigma(ps, entry) -> device pointer to {sigma, 1/sigma} of that entry (NULL if not normalised);
 *   sgv_pset_step: clip_grad_norm_(max_norm) (max_norm <= 0: no clipping) + AdamW(lr, betas (0.9, 0.999), eps 1e-8,
 *     weight_decay) on every entry; total_norm_host (may be NULL: no host sync) receives the norm before clipping. */
typedef struct sgv_pset sgv_pset;
typedef struct sgv_pset_entry { float* p; float* g; long n; int rows, cols; float* u; float* v; } sgv_pset_entry;
int sgv_pset_create(const sgv_pset_entry* entries, int n, sgv_pset** out);
int sgv_pset_destroy(sgv_pset* ps);
int sgv_pset_power_iteration(sgv_pset* ps, int train, void* stream);
const float* sgv_pset_sigma(const sgv_pset* ps, int entry);
int sgv_pset_step(sgv_pset* ps, float lr, float weight_decay, float max_norm, float* total_norm_host, void* stream);

/* [Bn][I][J] -> [Bn][J][I] with dtype conversion (reference NCHW fp32 <-> channels-last compute dtype). */
int sgv_op_transpose(int src_dtype, int dst_dtype, const void* src, void* dst, int Bn, int I, int J, void* stream);

#ifdef __cplusplus
}
#endif
#endif
