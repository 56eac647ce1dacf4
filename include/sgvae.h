/*
 * sgvae.h -- C ABI of libsgvae.so: the MI355X-native (gfx950) engine for the SimulGen-VAE
 * training step.
 *
 * The reference (leesihun/SimulGen-VAE) has no FFI/plugin boundary for this path: it sits behind
 * Python classes.  Each entry point below names the reference interface it stands in for
 * (paths relative to the reference root).  The build's own host-side mirror
 * (simulgen-vae_amd/modules/, same names and argument meaning as the reference's
 * modules/ package) binds these symbols through ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions: every function returns 0 on success or a negative sgv_status; the message of the
 * last failure on the calling thread is sgv_last_error().  No C++ exception crosses the ABI.
 * The caller owns every buffer it passes; the engine owns parameters, optimizer state and
 * workspace.  All device work is enqueued on the hipStream_t given to sgv_create (pass the
 * stream PyTorch-ROCm is using so that it composes with torch tensors); nothing here
 * synchronises the stream except the explicitly host-returning calls (marked [sync]).
 * One engine per process/thread; an engine is not thread-safe.
 *
 * Layouts: "reference layout" means exactly what the reference's tensors hold:
 * activations [B, C, T] row-major fp32, parameters as in its state_dict.  Internally the engine
 * keeps activations channels-last [B, T, C] and weights [tap][Cout][Cin]; conversion happens
 * inside these calls.
 */
#ifndef SGVAE_H
#define SGVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgv_engine sgv_engine;

typedef enum {
    SGV_OK = 0,
    SGV_ERR_ARG = -1,      /* bad argument / unsupported configuration */
    SGV_ERR_HIP = -2,      /* a HIP runtime call failed */
    SGV_ERR_STATE = -3,    /* call order (e.g. backward before forward) */
    SGV_ERR_NAME = -4,     /* unknown state_dict key */
    SGV_ERR_NOGPU = -5     /* no gfx950 device visible: there is no CPU fallback */
} sgv_status;

enum { SGV_DTYPE_F32 = 0, SGV_DTYPE_BF16 = 1 };
enum { SGV_LOSS_MSE = 0, SGV_LOSS_MAE = 1, SGV_LOSS_SMOOTHL1 = 2, SGV_LOSS_HUBER = 3 };
enum { SGV_MAX_LEVELS = 8 };

/* Mirrors the constructor arguments of modules.VAE_network.VAE (modules/VAE_network.py:60):
 * VAE(latent_dim, hierarchical_dim, num_filter_enc, num_filter_dec, num_node, num_time,
 *     lossfun, batch_size, small).  num_filter_dec is num_filter_enc reversed
 * (SimulGen-VAE.py:219) and is not passed separately. */
typedef struct {
    int32_t latent_dim;
    int32_t hierarchical_dim;
    int32_t n_levels;
    int32_t num_filter_enc[SGV_MAX_LEVELS];
    int32_t num_node;
    int32_t num_time;
    int32_t max_batch;        /* per-GPU batch the workspace is sized for */
    int32_t loss_type;        /* SGV_LOSS_* <- condition.txt Loss_type (SimulGen-VAE.py:208-215) */
    int32_t small;            /* 1 <=> --size small */
    int32_t compute_dtype;    /* SGV_DTYPE_*: storage/MFMA input type of activations and weight copies */
    int32_t flags;            /* bit0: TN GEMM uses scalar LDS fragment reads instead of ds_read_tr */
} sgv_config;

/* Scalars written by forward, in the order VAE.forward returns them (VAE_network.py:117):
 * [0] recon_loss (selected loss fn), [1] kl, [2] kl_2 stage 0, [3] kl_2 stage 1, ...,
 * [1+n_kl] recon_loss_MSE; n_kl = n_levels - 1. */
enum { SGV_MAX_SCALARS = 12 };

const char* sgv_last_error(void);

/* modules/VAE_network.py:60 VAE.__init__ (+ train.py:65,83 model creation / .to(device)). */
int sgv_create(const sgv_config* cfg, void* hip_stream, sgv_engine** out);
int sgv_destroy(sgv_engine* e);

/* nn.Module.state_dict() surface: key names/shapes of the reference (checkpoint row).
 * kind: 0 bias, 1 weight_orig, 2 weight_u, 3 weight_v, 4 GroupNorm weight, 5 GroupNorm bias. */
int sgv_param_count(const sgv_engine* e);
int sgv_param_info(const sgv_engine* e, int index, const char** name, int* ndim, int64_t shape[4],
                   int* kind, int* has_grad);
/* load_state_dict / state_dict(): host fp32 buffers in reference layout.  [sync] */
int sgv_load_state(sgv_engine* e, const char* name, const float* host, size_t count);
int sgv_export_state(sgv_engine* e, const char* name, float* host, size_t count);
/* p.grad of a parameter after sgv_backward (wrt weight_orig, i.e. with the spectral-norm chain
 * rule applied), reference layout; *is_none = 1 for parameters that never get a gradient
 * (train.py:157 `if p.grad is not None`).  [sync] */
int sgv_export_grad(sgv_engine* e, const char* name, float* host, size_t count, int* is_none);
/* Adam exp_avg / exp_avg_sq of a parameter (torch.optim.AdamW state), reference layout. [sync] */
int sgv_export_adam(sgv_engine* e, const char* name, float* host_m, float* host_v, size_t count);

/* Refresh the compute-dtype weight copies from the fp32 masters after sgv_load_state. */
int sgv_prepare(sgv_engine* e);

/* Batch input.  x_dev: device fp32 [batch, num_node, num_time] (the tensor train.py:142 hands to
 * model(image)); converted to the internal layout/dtype. */
int sgv_set_input(sgv_engine* e, const float* x_dev, int batch);
/* Reparameterisation noise (decoder.py:221 torch.randn_like), device fp32, reference layouts
 * [B,latent], [B,C,T]...; site 0 = top latent, 1.. = decoder stages.  If a site is never set the
 * engine draws it from its Philox stream (sgv_seed). */
int sgv_set_eps(sgv_engine* e, int site, const float* eps_dev, int batch);
int sgv_seed(sgv_engine* e, uint64_t seed);
/* Data-parallel noise ("Philox streams keyed by global sample index so results are world-size-invariant"): sample b
 * of this engine's batch is sample b * world + rank of the global batch (shards are r::world of the shuffled index list).  The
 * engine's own draws then take the Philox counters of that global row, so N ranks with the SAME seed and B / N samples each draw
 * exactly the noise one rank draws for B samples.  Default (0, 1). */
int sgv_set_shard(sgv_engine* e, int rank, int world);
/* Engine switches: "write_xhat" (materialise the reconstruction in training forwards; default 1),
 * "use_tr" (weight-gradient GEMM reads LDS with ds_read_b64_tr_b16; default 1), "dw_side_stream" (small weight-gradient
 * GEMMs on a second stream; default 1), "vendor_gemm" (removed in round 3: 0 is accepted, 1 is an error -- every GEMM
 * runs on the hand-written kernels; the hipBLASLt comparator lives in tests/micro/vendor, outside this library), "deterministic" (default 1: no
 * floating-point atomics anywhere in the step), "lanes" (second compute lane for the posterior branch of a decoder stage and
 * the xs heads; schedule only, results are bitwise the same; default 1), "fused_stages" (small Conv1d -> GroupNorm -> GELU
 * stages in one launch, csrc/convgn.hip; default 1, 0: GEMM + split-K combine + GroupNorm kernels),
 * "grad_bf16" (bf16 engines, single-GPU path; default 0, modules/train.py switches it on at world size 1: the weight gradients of
 * the layers whose weight-gradient GEMM is the 256 x 256 kernel reach the optimizer as bf16 -- rounded to nearest even in the
 * GEMM's epilogue, the rounding the data-parallel step's bf16 wire format applies to every weight gradient -- 4 bytes less written
 * and read per parameter; sgv_export_grad / sgv_grad_norm refresh the fp32 arena from the bf16 copy on demand; ignored while a
 * communicator or a bucket callback is registered). */
int sgv_set_option(sgv_engine* e, const char* key, int value);

/* VAE.forward (VAE_network.py:79-121) on the current input.  train != 0: spectral-norm power
 * iteration runs (model.train()); mode_fix != 0: Decoder.forward(mode="fix") (decoder.py:209-210).
 * scalars_host (may be NULL): SGV_MAX_SCALARS floats, filled after a stream sync.  [sync if given] */
int sgv_forward(sgv_engine* e, int train, int mode_fix, float* scalars_host);
/* Decoder.forward(z, xs, mode) from caller-supplied latents (utils.py:499, latent_conditioner_e2e.py:371,
 * reconstruction_evaluator.py:174): z_dev fp32 [B,latent], xs_dev fp32 [n_levels-1][B,hier] in the list
 * order Encoder.forward returns; eval-mode spectral norm.  Scalars as sgv_forward (the loss entries compare
 * against whatever input is current and are meaningless without one).  Result via sgv_get_xhat. */
int sgv_decode(sgv_engine* e, const float* z_dev, const float* xs_dev, int batch, int mode_fix, float* scalars_host);
/* Encoder.forward only (utils.py:492): mu, log_var [B,latent], xs [n_levels-1][B,hier] to host. [sync] */
int sgv_encode(sgv_engine* e, float* mu_host, float* logvar_host, float* xs_host);
/* Reconstruction of the last forward, reference layout [B, num_node, num_time] fp32 on device. */
int sgv_get_xhat(sgv_engine* e, float* xhat_dev);
/* Named intermediate of the last forward as fp32 reference layout on host (parity tests):
 * "enc_h<i>", "dec_out<i>", "zmap<i>", "mu", "log_var", "z", "xs<i>", "x_hat"; "x_in" = the input batch as
 * held by the engine after sgv_set_input / sgv_augment_collate (readable before any forward).  [sync] */
int sgv_get_activation(sgv_engine* e, const char* name, float* host, size_t count);

/* loss = alpha*recon + beta*sum(kl); loss.backward() (train.py:144-153). */
int sgv_backward(sgv_engine* e, float alpha, float beta);
/* sgv_backward followed by sgv_adamw_step(lr) (train.py:153-168 without the logging in between), single-GPU path:
 * the AdamW pass of each conv-weight bucket is started on a second stream as soon as that bucket's gradients are
 * final and runs under the remaining backward kernels (HBM-bound optimizer next to MFMA-bound GEMMs).  Same
 * results as the two separate calls; gradients stay readable (sgv_export_grad) afterwards.  Not valid while a
 * bucket callback is registered. */
int sgv_backward_step(sgv_engine* e, float alpha, float beta, float lr);
/* Callback invoked from inside sgv_backward (host side, after the kernels producing a gradient
 * bucket have been enqueued) so the caller can overlap its all-reduce of
 * [sgv_grad_buffer + offset, +count) with the rest of backward.  Weight buckets arrive in
 * reverse-autograd order; the small last-numbered bucket (biases, GroupNorm affine, <G,W> scalars) is released
 * just before the first encoder layer's weight bucket, which is always the final callback. */
typedef void (*sgv_bucket_cb)(void* user, int bucket, size_t offset_elems, size_t count_elems);
int sgv_set_bucket_callback(sgv_engine* e, sgv_bucket_cb cb, void* user);
/* The flat fp32 gradient arena (device) of all parameters that receive gradients; what the
 * data-parallel all-reduce (RCCL via torch.distributed) operates on. */
int sgv_grad_buffer(sgv_engine* e, float** dev_ptr, size_t* count_elems);
int sgv_scale_grads(sgv_engine* e, float factor);

/* Native RCCL path (sgv_allreduce_grads(rcclComm_t); 8(e): one collective per step, overlapped with the
 * decoder backward).  RCCL is resolved at run time (dlopen of librccl.so.1, i.e. the copy already loaded in the process
 * if there is one), so the library has no link-time dependency on it.
 *  - sgv_rccl_unique_id / sgv_rccl_comm_init / sgv_rccl_comm_destroy: ncclGetUniqueId (128 bytes, rank 0; the caller
 *    broadcasts them), ncclCommInitRank, ncclCommDestroy -- so a host needs no RCCL binding of its own.
 *  - sgv_allreduce_grads: mean all-reduce (ncclAvg, fp32) of the whole gradient arena, bucket by bucket in backward
 *    order, on comm_stream after everything enqueued so far on the engine stream; the engine stream then waits for it.
 *  - sgv_set_rccl: register (comm, comm_stream) with the engine (NULL comm unregisters).  While registered,
 *    sgv_backward issues each bucket's all-reduce itself as soon as the kernels producing it are enqueued (no host
 *    callback), sgv_adamw_step / sgv_adamw_step_range make the engine stream wait for exactly the buckets they touch, and
 *    sgv_adamw_step updates every layer whose bucket has arrived while the last (first-encoder-layer) bucket is in flight;
 *    sgv_backward_step = sgv_backward + that sgv_adamw_step.  Mutually exclusive with sgv_set_bucket_callback. */
int sgv_rccl_unique_id(void* id128);
/* Local checks a host runs BEFORE it enters any collective step of the set-up, so that every rank can agree (through its own
 * process group) on the path it takes: sgv_rccl_probe resolves librccl and its entry points in this process (no communication);
 * sgv_rccl_comm_count = ncclCommCount of a communicator made by sgv_rccl_comm_init (what the bench line reports as rccl_nranks). */
int sgv_rccl_probe(void);
int sgv_rccl_comm_count(void* comm, int* nranks);
/* In-place mean all-reduce (ncclAvg) of `count` fp32 / bf16 elements on `stream`: the host's first collective on a new
 * communicator (a bounded self-test before the first training step), and a building block for hosts without an RCCL binding. */
int sgv_rccl_allreduce(void* comm, void* dev_buf, size_t count, int dtype, void* stream);
int sgv_rccl_comm_init(void** comm_out, int nranks, const void* id128, int rank);
int sgv_rccl_comm_destroy(void* comm);
int sgv_allreduce_grads(sgv_engine* e, void* rccl_comm, void* comm_stream);
int sgv_set_rccl(sgv_engine* e, void* rccl_comm, void* comm_stream);
/* A communication stream owned by the engine for sgv_set_rccl, chosen so that it shares its hardware queue with neither the engine
 * stream nor the engine's weight-gradient / second-lane streams (the HIP runtime maps streams onto four hardware queues; kernels of
 * two streams on one queue never overlap, and a collective on the engine stream's queue would serialise with backward). */
int sgv_comm_stream(sgv_engine* e, void** hip_stream);

/* Gradient 2-norm as train.py:156-161 computes it.  [sync] */
int sgv_grad_norm(sgv_engine* e, double* out);
/* torch.optim.AdamW(lr).step() with its defaults (train.py:92,168): betas (0.9, 0.999),
 * eps 1e-8, weight_decay 0.01; skips parameters without gradient.  Also refreshes the
 * compute-dtype weight copies. */
int sgv_adamw_step(sgv_engine* e, float lr);
/* The same step restricted to the parameters whose gradients live in buckets [bucket_lo, bucket_hi) (bucket
 * numbering of sgv_set_bucket_callback; the last bucket holds biases, GroupNorm affine and the spectral-norm
 * <G,W> scalars, which every conv weight's update needs).  Lets a data-parallel caller update the layers whose
 * all-reduce has finished while the last, largest bucket is still in flight.  first=1 on the first call of an
 * optimisation step, last=1 on the final one; every bucket must be covered exactly once per step. */
int sgv_adamw_step_range(sgv_engine* e, float lr, int bucket_lo, int bucket_hi, int first, int last);
int sgv_bucket_count(const sgv_engine* e);
/* Optimizer overlap for the data-parallel step (the reference's loop is optimizer.step() after backward, modules/train.py:153-168;
 * torch's DDP hides the all-reduce under backward, this also hides the update).  The <G,W> scalars of the conv layers of weight
 * bucket b sit together at the head of the small bucket's range: sgv_bucket_dots returns that sub-range, and they are final when
 * the bucket's callback fires.  A caller that averages [offset, +count) together with bucket b -- and leaves the union of those
 * sub-ranges (the first sum-of-counts elements of the small bucket) out of the small bucket's collective -- may then call
 * sgv_adamw_bucket_async(lr, b) as soon as it has made the optimizer stream (sgv_opt_stream) wait for both collectives: the
 * bucket's conv weights are updated on that stream under the rest of backward.  The sgv_adamw_step_range calls that close the
 * step (same bucket coverage as without the overlap) skip what was updated ahead, and last=1 joins the optimizer stream.
 * With sgv_set_rccl, sgv_backward_step does all of this by itself (option "ddp_early_adamw", default 1).  Linear-head weights,
 * biases and GroupNorm affine are updated by the closing calls: their <G,W> scalars are computed at the end of backward. */
int sgv_bucket_dots(const sgv_engine* e, int bucket, size_t* offset_elems, size_t* count_elems);
/* Where a released bucket is complete.  By default on the engine stream: before the callback the engine stream joins the
 * weight-gradient side stream and, for the bf16 wire format, runs the pack pass.  With sgv_set_option("wire_stream", 1) both
 * happen on the wire stream instead (sgv_wire_stream): the callback must issue its collective ordered after THAT stream, and
 * backward never waits for the side stream or the pack pass at a release point. */
int sgv_wire_stream(sgv_engine* e, void** hip_stream);
int sgv_opt_stream(sgv_engine* e, void** hip_stream);
int sgv_adamw_bucket_async(sgv_engine* e, float lr, int bucket);
/* Wire format of the data-parallel gradient exchange (the reference's DDP all-reduces fp32 gradients, modules/utils.py:209-238
 * sets the process group up and torch does the rest; this is the build's own choice for bf16 engines).  SGV_DTYPE_F32: the
 * buckets are ranges of the fp32 arena (default).  SGV_DTYPE_BF16: at its fire point every weight bucket is rounded (RNE) into a
 * bf16 copy at the same element offset of the payload buffer; the collective averages THAT range (the bucket callback receives
 * the same offset / count; with sgv_set_rccl the engine issues ncclAllReduce on it), and sgv_adamw_step / _range write it back
 * into the fp32 arena in front of the bucket's update.  The last (small) bucket -- biases, GroupNorm affine, <G,W> scalars --
 * always travels in fp32.  Halves the bytes on the xGMI links (0.80 of 1.61 GB per step for preset 1).
 * sgv_grad_payload_unpack writes back every bucket still packed (use before reading gradients with sgv_grad_norm / sgv_get_grad
 * when no optimiser step follows). */
int sgv_set_grad_payload(sgv_engine* e, int dtype);
int sgv_grad_payload_buffer(sgv_engine* e, void** dev_ptr, size_t* count_elems);
int sgv_grad_payload_unpack(sgv_engine* e);
/* Device memory held by the engine, bytes: out[0] fp32 master parameters, [1] gradient arena, [2] Adam moments, [3] compute-dtype
 * weight copies, [4] activations of forward + backward at max_batch (everything stays resident: the reference's
 * use_checkpointing flag is forced to False in its code, modules/VAE_network.py:68, and no recompute exists here either),
 * [5] split-K / reduction workspaces. */
int sgv_memory_info(const sgv_engine* e, size_t out[6]);
/* Measurement hook for BASELINE configs[3] ("+ grad-checkpoint"; the reference forces use_checkpointing to False,
 * modules/VAE_network.py:68).  With sgv_set_option("recompute_activations", 1) the backward pass regenerates every stage's
 * GroupNorm + GELU output from the stored pre-normalisation map right before the stage's backward (one extra streaming pass per
 * stage; same values, buffers stay allocated): it TIMES what recompute would cost; sgv_recompute_bytes returns the bytes of
 * the maps the last backward regenerated = what a recompute build would not keep resident. */
int sgv_recompute_bytes(const sgv_engine* e, size_t* bytes);
/* Gradient 2-norm accumulated by the AdamW pass(es) of the current step (same value sgv_grad_norm computes in a
 * separate pass).  [sync] */
int sgv_last_grad_norm(sgv_engine* e, double* out);
/* Epoch statistics without a host round trip per step (the reference's loop reads .item() scalars after every step,
 * modules/train.py:156-161,171-174).  sgv_scalars_accumulate: enqueue "add the scalars of the step that has just been
 * enqueued (forward + optimizer) to the device-side accumulator"; sgv_scalars_read: the accumulator, host16 =
 * [sum recon, sum kl, sum kl2_0.., ..., [8] sum mse, [9] sum of gradient norms, [10] steps]; reset != 0 clears it. */
int sgv_scalars_accumulate(sgv_engine* e);
int sgv_scalars_read(sgv_engine* e, double* host16, int reset);

/* AugmentedDataset.__getitem__ x batch + default collate (augmentation.py:43-124) on a dataset
 * resident in HBM in the engine's internal layout: builds the input batch directly.
 * dataset_dev: [P][num_time][num_node] in compute dtype (see sgv_dataset_convert).
 * idx[b]: sample index; noise_seed[b]: 0 = no noise else Philox key; scale[b]: 1.0 = none;
 * mix_idx[b]: -1 = none; lam[b]: mixup weight (already clipped to [0.1,0.9]). */
int sgv_augment_collate(sgv_engine* e, const void* dataset_dev, int batch, const int32_t* idx,
                        const uint64_t* noise_seed, const float* scale, const int32_t* mix_idx,
                        const float* lam);
/* The same batch construction, prefetched (the reference's DataLoader builds batch i + 1 in worker processes while step i
 * runs, modules/data_processing.py + torch DataLoader prefetch; here the GPU builds it beside the step):
 * sgv_augment_stage enqueues the augmentation of the NEXT batch into a second input buffer on a stream of its own -- its
 * kernels are launched from inside the next forward pass, behind the first encoder layer -- and returns at once;
 * sgv_augment_advance makes the staged batch the current input (what sgv_augment_collate does in one call).  Staging again
 * before advancing replaces the staged batch.  The current batch, its forward / backward and sgv_set_input are untouched by a
 * staged one.  sgv_augment_advance without a staged batch: SGV_ERR_STATE.  Same bytes as sgv_augment_collate for the same
 * arguments.  Loop:  stage(0); for i: advance(); stage(i + 1); forward; backward_step. */
int sgv_augment_stage(sgv_engine* e, const void* dataset_dev, int batch, const int32_t* idx,
                      const uint64_t* noise_seed, const float* scale, const int32_t* mix_idx,
                      const float* lam);
int sgv_augment_advance(sgv_engine* e);
/* utils.Dataset.__init__ with load_all (utils.py:41-43): device fp32 [count, num_node, num_time]
 * -> internal [count][num_time][num_node] in compute dtype at dst_dev. */
int sgv_dataset_convert(sgv_engine* e, const float* src_dev, void* dst_dev, int count);
size_t sgv_dataset_sample_bytes(const sgv_engine* e);

/* Input pipeline, stateless.  data_preprocess.data_scaler (data_preprocess.py:65-165) fits
 * sklearn.preprocessing.MinMaxScaler(feature_range=(-0.7, 0.7)) per node on a sample of [num_time x num_param]
 * rows and transforms the whole [P][T][N] array; SimulGen-VAE.py:282 then transposes it for Conv1d.
 * sgv_minmax_fit: per-node min / max (NaNs ignored) over n_rows device rows of n_node floats (accumulate=1 merges
 * into existing values, for chunked feeding) [sync]; sgv_minmax_coeffs: MinMaxScaler's scale_ / min_ (zero ranges
 * handled as sklearn does); sgv_scale_convert: x*scale_+min_ of raw rows written in the engine's resident
 * dataset layout [P][T][N] and compute dtype (raw [P][T][N] rows are already in that order). */
int sgv_minmax_fit(const float* rows_dev, long n_rows, int n_node, float* min_dev, float* max_dev, int accumulate,
                   void* stream);
int sgv_minmax_coeffs(const float* min_dev, const float* max_dev, int n_node, float lo, float hi, float* scale_dev,
                      float* offset_dev, void* stream);
int sgv_scale_convert(int dst_dtype, const float* src_dev, const float* scale_dev, const float* offset_dev,
                      void* dst_dev, long n_rows, int n_node, void* stream);

/* Profiling aid for bench.py: hipEvent-timed duration (total ms and number of launches since the
 * last reset) of a kernel class ("gemm_nt", "gemm_tn"), measured on the engine's own stream. [sync] */
int sgv_kernel_time(sgv_engine* e, const char* which, float* total_ms, int* calls);
int sgv_kernel_time_reset(sgv_engine* e, int enable);   /* 0 off, 1 per class, 2 per (class, layer, shape) */
/* Walk the timer tags recorded since the engine was created: index 0.. until the return value is
 * non-zero.  name receives "class" or "class|layer prefix|M=.. N=.. K=.. taps=.. splitk=..". [sync] */
int sgv_kernel_time_tag(sgv_engine* e, int index, char* name, size_t cap, float* total_ms, int* calls);

/* Low-level kernel entry points, exported for the unit parity tests (tests/test_kernels_gpu.py).
 * All pointers are device pointers; dtype is SGV_DTYPE_*.  [sync] */
int sgv_test_gemm_nt(int dtype, const void* A, const void* W, void* C, const float* bias, const float* scale,
                     const void* addend, int M, int N, int K, int taps, int Tlen, int splitk, int out_f32,
                     void* stream);
/* bf16 128x128 kernel with the GroupNorm-statistics epilogue: sums[(m / Tlen) * (N / Cg) + n / Cg][2] += (sum, sum of squares)
 * of the stored outputs; sums must be zeroed by the caller.  Rejected unless Tlen >= 128, Cg >= 128 and the shape runs on the
 * 128x128 kernel (fewer than 64 K-steps of 32, or N < 256). */
int sgv_test_gemm_nt_stats(const void* A, const void* W, void* C, const float* bias, const void* addend, int M, int N, int K,
                           int taps, int Tlen, int Cg, double* sums, void* stream);
/* Test hook: the fused small Conv1d + GroupNorm(G) + GELU (+ residual) forward kernel (csrc/convgn.hip) on caller-owned device
 * buffers: A [B*T][K] bf16, W [taps][N][K] bf16, y / out / res [B*T][N] bf16, sums [B*G][2] doubles.  Fails for shapes the kernel
 * does not take (T > 208, channels per group not in {16, 32, 64, 80, 128, 160}, K % 32 != 0). */
int sgv_test_conv_gn_fwd(const void* A, const void* W, const float* bias, const float* scale, const void* res, const float* gamma,
                         const float* beta, void* y, void* out, double* sums, int B, int T, int N, int K, int taps, int G, float rscale,
                         void* stream);
/* Test hook: its backward mirror -- input gradient of the upper convolution (A = its dY [B*T][K], W = its transposed tap-flipped
 * weight copy [taps][N][K], optional addend [B*T][N] bf16 added before rounding: the residual path; optional premul x [B*T][N]: the
 * gradient is multiplied by gelu'(x) and rounded again; optional da [B*T][N]: the input gradient itself is stored too) + GroupNorm / GELU backward
 * of the stage below (y, forward sums, gamma, beta, conv bias) in one
 * launch: dy [B*T][N] bf16, sums2 [B*G][2], ptot [B][3][N] (per-sample column totals), cdot_part [B*G]. */
int sgv_test_conv_gn_bwd(const void* A, const void* W, const float* scale, const void* addend, const void* premul, void* da, const void* y,
                         const double* sums, const float* gamma,
                         const float* beta, const float* cbias, void* dy, double* sums2, float* ptot, float* cdot_part, int B, int T,
                         int N, int K, int taps, int G, void* stream);
/* Test hook for the 256x256 persistent implicit-GEMM kernel (csrc/gemm256.hip; replaces the hipBLASLt dispatch of round 1 on
 * the reference call sites modules/decoder.py:117-121, modules/common.py:135-141, modules/encoder.py:34).  bf16 operands.
 * mode 0: the kernel itself with the given split-K; mode 1: the engine's kernel choice (gemm_nt_plan: 256x256 kernel, or
 * 256x256 over the first rows + the 128-row kernels over the rest, or the 128-row kernels alone); *plan_kind reports it.
 * sums != NULL: GroupNorm (sum, sum of squares) per (sample, group of Cg channels), [ceil(M/Tlen)][N/Cg][2] fp64, produced
 * deterministically by the epilogue + a fixed-order finalize.
 * Bits 8-15 of mode choose the work-item order of the 256x256 kernel (0: the launcher's choice, 1..254: row tiles in bands of
 * that many, 255: column-panel-major), bits 16-18 the cache policy of its streams (0: the launcher's choice, 7: default loads and
 * stores, else bit 0 = non-temporal weight loads, bit 1 = sc1 output stores), bits 19-20 its tile shape (0: 256 x 256 with mode 0 /
 * the plan's choice with mode 1, 1: 128 x 512, 2: the plan must not pick 128 x 512), bit 21 the split the engine runs on two
 * streams (M = 128 mod 256, bf16 output: the first M - 128 rows on the 256 x 256 kernel in `splitk` slices, the last 128 rows as a
 * launch of their own of the 128 x 512 tile shape on shifted row pointers; here one after the other on `stream`). */
int sgv_test_gemm_nt256(const void* A, const void* W, void* C, const float* bias, const float* scale, const void* addend, int M, int N,
                        int K, int taps, int Tlen, int splitk, int out_f32, int mode, int Cg, double* sums, int* plan_kind, void* stream);
/* Weight-gradient GEMM test hook.  use_tr: 0 plain LDS reads, 1 the launcher's choice, 2 force the 128x256 two-blocks-per-CU kernel,
 * 3 the same with its persistent walk, 4 force the persistent 256x256 kernel (csrc/gemm256tn.hip), 5 never that kernel, 6 that kernel
 * with its bf16 epilogue (dW is then a bf16 array; splitk 1), 7 that kernel in its work-stealing form (what a data-parallel backward
 * launches while a collective may hold CUs; results bitwise those of 4). */
int sgv_test_gemm_tn(int dtype, const void* A, const void* Bm, float* dW, int M, int N1, int N2, int taps,
                     int Tlen, int splitk, int use_tr, void* stream);
/* Test hook for the stream placement: *overlaps = 1 if kernels of the engine's auxiliary stream `which` (0 second lane, 1 weight-
 * gradient side stream, 2 optimizer stream, 3 the engine's communication stream; 2 and 3 are created by the call if need be) can run
 * while a kernel of the engine stream is running, i.e. the two do not share a hardware queue; -1 if the engine has no such stream. */
int sgv_test_stream_overlap(sgv_engine* e, int which, int* overlaps);
/* Test hook: `blocks` workgroups of `threads` threads and `lds_bytes` of LDS spin for `ticks` of the 100 MHz clock (<= 10 ms) on
 * `stream` -- a stand-in for a collective's resident channel workgroups when measuring how the persistent GEMMs cope. */
int sgv_test_occupy(void* stream, int blocks, int threads, int lds_bytes, long long ticks);
/* Test double for the collective of the engine-issued data-parallel step (sgv_set_rccl with any non-null communicator handle):
 * k != 0 replaces ncclAllReduce by "multiply the range in place by k on the given stream" (fp32 and bf16 ranges; no RCCL is
 * loaded or called), k == 0 restores the library.  With k a power of two a step through the double must leave bitwise the
 * state of a plain step at (k alpha, k beta) if and only if every gradient element and every <G,W> slot went through exactly
 * one collective.  *calls / *elems return (and reset) the number of collectives and of elements since the last call. */
int sgv_test_fake_collective(float k, long* calls, long* elems);

#ifdef __cplusplus
}
#endif
#endif /* SGVAE_H */
