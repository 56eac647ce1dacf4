// libsgvae engine: parameter/optimizer arenas, layer graph of the hierarchical VAE, forward /
// backward orchestration on one HIP stream, and the C ABI of include/sgvae.h.
//
// Graph restated from the reference (channels-last, weights [tap][Cout][Cin]):
//   VAE.forward modules/VAE_network.py:79-121 ; Encoder modules/encoder.py:96-167 ;
//   Decoder modules/decoder.py:84-223 ; blocks modules/common.py:78-162 ; losses modules/losses.py:8-48 ;
//   training step modules/train.py:139-168.
// The backward pass is written out by hand (the reference uses autograd).
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/sgvae.h"
#include "sgv_ew.h"

static thread_local char g_err[1024] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
// same, for the other translation units of the library (cnn.hip)
int sgv_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(SGV_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHK(x)                                  \
    do {                                        \
        int r_ = (x);                           \
        if (r_ != 0) return r_ < 0 ? r_ : -r_;  \
    } while (0)

static const size_t NPOS = (size_t)-1;

struct Tensor {
    void* p = nullptr;
    int C = 0;
    long ld = 0;
    bool f32 = false;
};

enum { OP_CONV = 0, OP_CONVT = 1, OP_LINEAR = 2 };
enum { LIN_NONE = 0, LIN_HEAD = 1, LIN_EXPAND = 2 };

struct Layer {
    std::string prefix;
    int op = OP_CONV, cin = 0, cout = 0, k = 1;
    bool used = true, has_grad = true, need_wct = true;
    int lin_kind = LIN_NONE, lin_C = 0;      // head: K = lin_C*T; expand: O = lin_C*T
    size_t w = NPOS, b = NPOS, u = NPOS, v = NPOS;  // param arena (floats)
    size_t gw = NPOS, gb = NPOS, gdot = NPOS;       // grad arena (floats); gdot: <G,W_eff> scalar (small zone)
    size_t wc = NPOS, wct = NPOS;                   // compute-copy arena (elements)
    int sn = -1;
    int splitk_tn = 1;
    size_t dot_part = NPOS;                         // per-block <G,W_eff> partials of the layer's dY kernel (e->red arena)
    size_t col_part = NPOS;                         // per-block column sums of dY (bias gradient of a conv without GroupNorm), same arena
    bool lp = false;                                // option grad_bf16: this layer's weight gradient lives in the bf16 mirror arena (fixed at creation:
                                                    //   its weight-gradient GEMM takes the 256 x 256 kernel at the engine's full batch)
    long nw() const { return (long)cout * cin * k; }
};
struct GNLayer {
    std::string prefix;
    int C = 0, G = 1;
    bool used = true, has_grad = true;
    size_t gamma = NPOS, beta = NPOS, ggamma = NPOS, gbeta = NPOS;
    size_t ptot = NPOS;                             // [B][3][C] per-sample column totals of the backward pass (e->red arena)
};
struct Stage {
    int layer = -1, gn = -1, act = 0;
    bool pre_gelu = false, out_f32 = false;
    Tensor pre, y, a, dy, da, dpre;
    size_t sums = NPOS, sums2 = NPOS;   // stats arena (doubles)
};
struct Block {
    std::vector<Stage> st;
    bool residual = false;
};
struct StateEntry {
    std::string name;
    int kind;  // 0 bias,1 weight_orig,2 u,3 v,4 gn w,5 gn b
    int layer = -1, gn = -1;
    std::vector<int64_t> shape;
    bool has_grad;
    long count() const { long n = 1; for (auto s : shape) n *= s; return n; }
};

struct TimerRec { hipEvent_t a, b; int tag; };

struct sgv_engine {
    sgv_config cfg;
    hipStream_t stream = nullptr;
    int dt = 0;         // compute dtype
    size_t esz = 4;     // bytes per compute element
    int n = 0, n_st = 0, T = 0, N = 0, Z = 0, H = 0, maxB = 0;
    std::vector<int> enc, dec;
    std::vector<Layer> layers;
    std::vector<GNLayer> gns;
    std::vector<StateEntry> entries;
    std::map<std::string, int> entry_index;
    // arenas
    float* params = nullptr; size_t n_params = 0;
    float* grads = nullptr; size_t n_grads = 0, n_grads_w = 0;   // weights zone first, small zone after
    float* adam_m = nullptr; float* adam_v = nullptr;
    char* copies = nullptr; size_t n_copies = 0;
    char* act = nullptr; size_t act_bytes = 0, act_used = 0;
    double* stats = nullptr; size_t n_stats = 0, n_stats_fwd = 0;  // [fwd sums | bwd sums2]
    float* sn_tmp = nullptr; size_t n_sn_tmp = 0;   // [tmp_t of fused layers][tmp_t of the others][tmp_s of all]
    size_t n_sn_tmp_fused = 0, sn_tmp_s_off = 0;
    bool wtu_fresh = false;                          // tpart of the fused layers holds the W^T u partials for the current weights
    std::vector<size_t> sn_tpart_off, sn_spart_off;  // per layer: offsets of the power-iteration partials inside sn_tmp
    WorkItem* items_ts = nullptr; WorkItem* items_ss = nullptr; int n_items_ts = 0, n_items_ss = 0;
    float* lin_dot_part = nullptr;                   // per-work-item <G,W>/sigma partials of the Linear layers
    std::vector<FinDot> fin_lin_dots;
    std::vector<int> dot_off, fin_lin_off;           // [bucket] -> first Linear <G,W> work item / first fin_lin_dots entry (tables sorted by bucket)
    double* gnorm_part = nullptr; int n_gnorm_part = 0;   // per-work-item sums of squared gradients of the AdamW passes
    float* sn_sigma = nullptr;
    float* sn_dot_dummy = nullptr;
    double* scal = nullptr;        // device doubles: [0..1] loss sums, [2] kl, [3..] kl2, [15] grad norm^2
    float* partial = nullptr; size_t partial_floats = 0;
    // weight-gradient GEMMs are off the critical path of backward: the SMALL ones (<= 250 GFLOP, i.e. everything but
    // the five largest layers) run on a second stream next to the dX GEMMs and normalisation passes of the following
    // layers, which fills the CUs those 100-200-block launches leave idle (measured 16.14 -> 15.83 ms/step).  Putting
    // the big ones there too loses 2.5 %: they fill every CU on their own and co-running kernels evict each other's
    // L2 tiles.  Option "dw_side_stream" / SGV_DW_SIDE=0 turns it off; kernel-timing passes always run on one stream.
    hipStream_t side = nullptr;
    std::vector<char> aug_host[4]; int aug_turn = 0;                  // staging of sgv_augment_collate's control arrays
    void* comm = nullptr; hipStream_t comm_stream = nullptr;          // native RCCL path (sgv_set_rccl)
    std::vector<hipEvent_t> bucket_done; std::vector<char> bucket_pending;
    // data-parallel wire format of the weight buckets: 0 = the fp32 arena itself, 1 = a bf16 copy (packed at the bucket's fire point,
    // averaged by the collective, unpacked into the arena in front of the bucket's AdamW).  The small bucket always travels in fp32.
    int payload_bf16 = 0; void* grads_lp = nullptr; std::vector<char> bucket_packed;
    // option "grad_bf16" (bf16 engines, single-GPU path: no communicator, no bucket callback): the 256 x 256 weight-gradient kernel
    // stores its result as bf16 into the mirror arena grads_lp and the AdamW pass reads it there (AdamDesc::glp) -- 4 B less
    // written and read per parameter of the big layers.  The fp32 arena of those layers is refreshed on demand (lp_sync) for the
    // calls that read it (sgv_export_grad, sgv_grad_norm).
    // With the bf16 wire format of the data-parallel step the same kernel writes the wire copy directly (bit for bit what the pack
    // pass produced from the fp32 result; that pass then skips those layers).
    int grad_bf16 = 0;
    bool lp_classified = false;
    std::vector<char> lp_dirty;          // per layer: its gradient of the last backward was stored as bf16 into grads_lp (not into the fp32 arena)
    std::vector<std::vector<int>> bucket_lp_layers;     // per weight bucket: its Layer::lp layers in arena order
    bool dw_chunk_direct = false;        // chunked first-layer gradient (data-parallel): the chunk GEMMs write the wire copy themselves
    // data-parallel optimizer overlap: the <G,W_eff> scalars of a weight bucket's layers sit together at the head of the small zone
    // (bucket_dots[b] = their range), so they can be averaged WITH the bucket instead of with the small bucket at the end of backward;
    // the bucket's conv-weight AdamW then runs on `opt` as soon as both collectives have landed, under the rest of backward
    // (sgv_adamw_bucket_async; the engine's own RCCL path does it by itself in sgv_backward_step).  bucket_updated[b]: done this step.
    std::vector<std::pair<size_t, size_t>> bucket_dots; size_t dots_total = 0;
    hipStream_t opt = nullptr; bool opt_dirty = false, adam_open = false;
    hipStream_t comm_own = nullptr;                   // sgv_comm_stream: a probed communication stream the engine owns
    hipStream_t wire = nullptr; int use_wire = 0;     // callback path: buckets are complete (and packed) on this stream, not on the engine stream
    std::vector<char> bucket_updated;
    int ddp_early = getenv("SGV_DDP_EARLY") ? atoi(getenv("SGV_DDP_EARLY")) : 1;
    // the last weight bucket (the first encoder layer: 97 M gradients that exist only when backward ends) is produced, exchanged and
    // updated in row chunks of the weight-gradient GEMM: chunk c's pack / all-reduce / AdamW run under chunk c + 1's GEMM, so only
    // the last chunk's exchange is exposed (engine-issued path; SGV_DDP_LAST_CHUNKS=1 turns it off).  Two chunks: 512 rows keep the
    // GEMM's 128 x 256 tiles at whole rounds of the chip, four chunks of 256 rows cost 27 % of the GEMM
    // BASELINE configs[3] "+ grad-checkpoint": what recomputing the GroupNorm + GELU outputs in backward would cost.  With the option on,
    // block_bwd regenerates every stage's activation a = act(GN(y)) from the stored pre-normalisation map and statistics right before
    // the stage's backward reads it (one extra streaming pass per stage).  The buffers themselves stay allocated -- this times the
    // recompute, it does not free the memory (sgv_memory_info's "activations" minus what recompute_bytes reports is what a
    // recompute build would keep); `use_checkpointing` stays forced off as in the reference
    bool recompute_act = false;
    size_t recompute_bytes = 0;
    int ddp_last_chunks = getenv("SGV_DDP_LAST_CHUNKS") ? atoi(getenv("SGV_DDP_LAST_CHUNKS")) : 2;
    double ddp_chunk_min_gf = getenv("SGV_DDP_CHUNK_MIN_GF") ? atof(getenv("SGV_DDP_CHUNK_MIN_GF")) : 250.0;   // tests lower it to chunk a small first layer
    int dw_chunks = 1, dw_chunk_layer = -1;
    std::function<int(int, int, int, int)> dw_chunk_hook;        // (chunk, chunks, first row, end row) after the chunk's GEMM is enqueued
    // bf16 wire format: the conv-weight AdamW reads a packed bucket straight from the averaged bf16 copy (no unpack pass; the fp32
    // arena keeps this rank's own gradients); only the few weights of a bucket that the flat pass updates (Linear heads:
    // bucket_flat_w) are unpacked.  bucket_packed[b]: bit 0 = conv-weight part still packed, bit 1 = flat part still packed.
    int lp_direct = getenv("SGV_LP_DIRECT") ? atoi(getenv("SGV_LP_DIRECT")) : 1;
    std::vector<std::vector<std::pair<size_t, size_t>>> bucket_flat_w;
    float* partial_tn = nullptr; size_t partial_tn_floats = 0;
    std::vector<hipEvent_t> ev_pool; size_t ev_next = 0;
    bool use_side = true, side_dirty = false;
    float* xpose_tmp = nullptr; size_t xpose_floats = 0;
    float* recon_unit = nullptr;   // [3][N] unit-scale dgamma/dbeta/dbias of the recon head
    float* colpart = nullptr; size_t colpart_floats = 0;   // per-block column-sum workspace
    SNDesc* sn_dev = nullptr; std::vector<SNDesc> sn_host;
    AdamDesc* adam_dev = nullptr; std::vector<AdamDesc> adam_host;
    WorkItem *items_sn = nullptr, *items_dot = nullptr, *items_adam = nullptr, *items_copy = nullptr, *items_wct = nullptr;
    WorkItem *items_sn_unf = nullptr, *items_adam_flat = nullptr, *items_adam_2d = nullptr;
    int n_items_sn = 0, n_items_dot = 0, n_items_adam = 0, n_items_copy = 0, n_items_wct = 0;
    int n_items_sn_unf = 0, n_items_adam_flat = 0, n_items_adam_2d = 0;
    std::vector<int> flat_off, tile_off;             // AdamW work items are sorted by gradient bucket: [bucket] -> first item
    // graph
    std::vector<Block> encA, encR, decU, decD, decP1, decP2, decX, decQ1, decQ2;
    Block decS, recon;
    std::vector<int> xs_lin, xs_exp;   // layer ids: encoder.xs_linear.i ; decoder.xs_sequence.i.0
    int last_lin = -1, start_lin = -1;
    // tensors
    Tensor x_in, xhat, sbuf, d_sbuf, dy_recon;
    // Prefetched augmentation (sgv_augment_stage / sgv_augment_advance): the NEXT batch is built in the spare input buffer on a
    // stream of its own, in launches of a few samples, beside the short kernels that follow the first encoder layer's GEMM (the
    // chip's HBM is idle there); the reference hides the same work in DataLoader worker processes.
    Tensor x_bufs[2];
    int x_cur = 0;
    hipStream_t aug_stream = nullptr;
    hipEvent_t aug_done = nullptr, aug_gate = nullptr, x_free[2] = {nullptr, nullptr};
    bool x_free_set[2] = {false, false};
    bool aug_staged = false, aug_fired = false, aug_pending = false;   // staged: control arrays on the device; fired: kernels enqueued; pending: the current batch's kernels may still run
    const void* aug_data = nullptr;
    int aug_next_batch = 0;
    char* aug_ctl = nullptr;           // control arrays of the staged batch
    std::vector<Tensor> enc_h, d_h, enc_a_dummy, zs, dzs, cat, dcat, dec_out, d_out, d_u, d_pres, d_qres, d_outp, gp, gq, xl, d_xl;
    std::vector<float*> xs_raw, d_xs_raw, eps, zmap;
    std::vector<int> eps_set;
    float *last = nullptr, *d_last = nullptr, *zlat = nullptr, *d_z = nullptr;
    int batch = 0;
    bool have_fwd = false, fwd_train = false, write_xhat = true, copies_fresh = false;
    int deterministic = getenv("SGV_DETERMINISTIC") ? atoi(getenv("SGV_DETERMINISTIC")) : 1;   // 1: no float-atomic accumulation anywhere in the step; option "deterministic"
    float* gn_part = nullptr; size_t gn_part_floats = 0;   // per-(tile, wave) GroupNorm partial sums of the 256x256 GEMM epilogue
    // deterministic reductions: block partials that nobody needs before the optimizer (GroupNorm affine / bias gradients,
    // <G,W_eff>) stay in this arena until the bucket they belong to is released, then two table-driven passes sum them
    // second compute lane: the posterior branch of a decoder stage (xs lift, condition_xz) is independent of the prior branch
    // (condition_z) between the residual block and the KL / reparameterisation kernel, in forward and in backward; both are chains
    // of small kernels that leave most of the chip idle, so they run side by side on two streams with workspaces of their own
    hipStream_t lane2 = nullptr; float* partial2 = nullptr; float* colpart2 = nullptr; float* gn_part2 = nullptr;
    hipEvent_t lane_fork = nullptr, lane_join = nullptr;
    hipEvent_t tail_fork = nullptr, tail_join = nullptr;      // concurrent 128-row tail of a 256 x 256 launch (launch_nt)
    bool coll_inflight = false;        // data-parallel backward, from the first released bucket on: a collective's channel workgroups may hold CUs
    int* tn_sched = nullptr;           // 8 x 520 ints: work-stealing state of the 256 x 256 weight-gradient launches issued while coll_inflight
    unsigned tn_sched_next = 0;
    int use_lanes = getenv("SGV_LANES") ? atoi(getenv("SGV_LANES")) : 1;
    // small Conv1d -> GroupNorm -> GELU stages in one launch (convgn.hip); SGV_CONVGN=0 restores GEMM + combine + GroupNorm kernels
    int use_convgn = getenv("SGV_CONVGN") ? atoi(getenv("SGV_CONVGN")) : 1;
    long convgn_maxk = getenv("SGV_CONVGN_MAXK") ? atol(getenv("SGV_CONVGN_MAXK")) : 4096;
    float* red = nullptr; size_t red_floats = 0;
    std::vector<FinDot> fin_dots; std::vector<FinAffine> fin_affine;
    int dot_counts[512];
    uint64_t seed = 0x5347564145ull, draw = 0;
    int shard_rank = 0, shard_world = 1;        // sgv_set_shard: sample b of this engine's batch is sample b * world + rank of the global batch
    long step = 0;
    float scalars_host[SGV_MAX_SCALARS];
    sgv_bucket_cb cb = nullptr; void* cb_user = nullptr;
    std::vector<std::pair<size_t, size_t>> buckets;   // (offset, count) in grad arena, backward order
    bool timing = false, timing_detail = false;
    std::vector<TimerRec> timers;
    std::map<std::string, int> tag_ids;
    std::vector<std::string> tag_names;
    int use_tr = 1;
};

// ------------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static int gn_groups(int c) { int g = c / 4; if (g < 1) g = 1; if (g > 8) g = 8; return g; }

struct Builder {
    sgv_engine* e;
    size_t np = 0, ngw = 0, ngs = 0, ncp = 0;
    std::vector<size_t*> small_grad_slots;   // gb / ggamma / gbeta offsets get rebased after the weight zone
    int add_layer(const std::string& prefix, int op, int cin, int cout, int k, bool used, bool has_grad, bool need_wct,
                  int lin_kind = LIN_NONE, int lin_C = 0) {
        Layer l;
        l.prefix = prefix; l.op = op; l.cin = cin; l.cout = cout; l.k = k;
        l.used = used; l.has_grad = has_grad; l.need_wct = need_wct && used && op != OP_LINEAR;
        l.lin_kind = lin_kind; l.lin_C = lin_C;
        e->layers.push_back(l);
        return (int)e->layers.size() - 1;
    }
    int add_gn(const std::string& prefix, int C, bool used, bool has_grad) {
        GNLayer g;
        g.prefix = prefix; g.C = C; g.G = gn_groups(C); g.used = used; g.has_grad = has_grad;
        e->gns.push_back(g);
        return (int)e->gns.size() - 1;
    }
};

static Tensor alloc_act(sgv_engine* e, long rows, int C, bool f32 = false) {
    Tensor t;
    t.C = C; t.ld = C; t.f32 = f32;
    size_t bytes = (size_t)rows * C * (f32 ? 4 : e->esz);
    e->act_used = align_up(e->act_used, 256);
    t.p = (void*)(e->act_used);   // offset for now; rebased after allocation
    e->act_used += bytes;
    return t;
}
static Tensor view_cols(const Tensor& t, int c0, int C, sgv_engine* e) {
    Tensor v = t;
    v.p = (char*)t.p + (size_t)c0 * (t.f32 ? 4 : e->esz);
    v.C = C;
    return v;
}

// ---- state-entry list in reference order (mirrors simulgen-vae_amd/spec.py) --------------------
static void add_entries_for_layer(sgv_engine* e, int li) {
    const Layer& l = e->layers[li];
    auto push = [&](const char* suffix, int kind, std::vector<int64_t> shape, bool hg) {
        StateEntry s;
        s.name = l.prefix + suffix; s.kind = kind; s.layer = li; s.shape = shape; s.has_grad = hg;
        e->entry_index[s.name] = (int)e->entries.size();
        e->entries.push_back(s);
    };
    push(".bias", 0, {l.cout}, l.has_grad);
    if (l.op == OP_CONV) push(".weight_orig", 1, {l.cout, l.cin, l.k}, l.has_grad);
    else if (l.op == OP_CONVT) push(".weight_orig", 1, {l.cin, l.cout, l.k}, l.has_grad);
    else push(".weight_orig", 1, {l.cout, l.cin}, l.has_grad);
    push(".weight_u", 2, {l.cout}, false);
    push(".weight_v", 3, {(int64_t)l.cin * l.k}, false);
}
static void add_entries_for_gn(sgv_engine* e, int gi) {
    const GNLayer& g = e->gns[gi];
    StateEntry s;
    s.name = g.prefix + ".weight"; s.kind = 4; s.gn = gi; s.shape = {g.C}; s.has_grad = g.has_grad;
    e->entry_index[s.name] = (int)e->entries.size(); e->entries.push_back(s);
    s.name = g.prefix + ".bias"; s.kind = 5;
    e->entry_index[s.name] = (int)e->entries.size(); e->entries.push_back(s);
}

static Stage mk_stage(int layer, int gn, int act, bool pre_gelu = false, bool out_f32 = false) {
    Stage s;
    s.layer = layer; s.gn = gn; s.act = act; s.pre_gelu = pre_gelu; s.out_f32 = out_f32;
    return s;
}

// Build layers + blocks in the reference's module registration order so that `entries` comes out in
// state_dict order (encoder: blocks, residual blocks, xs_linear, last; decoder: blocks, residual blocks,
// recon, sequence_start, xs_sequence, condition_z, condition_xz).
static int build_graph(sgv_engine* e) {
    Builder B{e};
    const bool small = e->cfg.small != 0;
    const int n = e->n, T = e->T;
    char buf[256];
    auto P = [&](const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap); return std::string(buf); };
    std::vector<std::function<void()>> entry_order;

    e->encA.resize(n); e->encR.resize(n);
    for (int i = 0; i < n; ++i) {
        const int cin = i == 0 ? e->N : e->enc[i - 1], C = e->enc[i];
        std::string p = P("encoder.encoder_blocks.%d.module_list.0._seq", i);
        int l0 = B.add_layer(p + ".0", OP_CONV, cin, C, 1, true, true, i > 0);
        int g0 = B.add_gn(p + ".1", C, true, true);
        add_entries_for_layer(e, l0); add_entries_for_gn(e, g0);
        e->encA[i].st.push_back(mk_stage(l0, g0, 1));
        if (!small) {
            int l1 = B.add_layer(p + ".3", OP_CONV, C, C, 3, true, true, true);
            int g1 = B.add_gn(p + ".4", C, true, true);
            add_entries_for_layer(e, l1); add_entries_for_gn(e, g1);
            e->encA[i].st.push_back(mk_stage(l1, g1, 1));
        }
    }
    for (int i = 0; i < n; ++i) {
        const int C = e->enc[i];
        std::string p = P("encoder.encoder_residual_blocks.%d.seq", i);
        e->encR[i].residual = true;
        for (int r = 0; r < (small ? 1 : 2); ++r) {
            int l = B.add_layer(p + P(".%d", r * 3), OP_CONV, C, C, 3, true, true, true);
            int g = B.add_gn(p + P(".%d", r * 3 + 1), C, true, true);
            add_entries_for_layer(e, l); add_entries_for_gn(e, g);
            e->encR[i].st.push_back(mk_stage(l, g, 1));
        }
    }
    for (int i = 0; i < n; ++i) {
        const bool dead = (i == 0) || (i == n - 1);
        int l = B.add_layer(P("encoder.xs_linear.%d", i), OP_LINEAR, e->enc[i] * T, e->H, 1, true, !dead, false, LIN_HEAD, e->enc[i]);
        add_entries_for_layer(e, l);
        e->xs_lin.push_back(l);
    }
    e->last_lin = B.add_layer("encoder.last_x_linear", OP_LINEAR, e->enc[n - 1] * T, 2 * e->Z, 1, true, true, false, LIN_HEAD, e->enc[n - 1]);
    add_entries_for_layer(e, e->last_lin);

    const int n_st = e->n_st;
    e->decU.resize(n_st); e->decD.resize(n_st);
    e->decP1.resize(n_st); e->decP2.resize(n_st); e->decX.resize(n_st); e->decQ1.resize(n_st); e->decQ2.resize(n_st);
    for (int i = 0; i < n_st; ++i) {
        int l = B.add_layer(P("decoder.decoder_blocks.%d.module_list.0._seq.0", i), OP_CONVT, e->dec[i], e->dec[i + 1], 3, true, true, true);
        add_entries_for_layer(e, l);
        e->decU[i].st.push_back(mk_stage(l, -1, 1));
    }
    for (int i = 0; i < n_st; ++i) {
        const int C = e->dec[i + 1];
        std::string p = P("decoder.decoder_residual_blocks.%d.seq", i);
        e->decD[i].residual = true;
        struct CS { int cin, cout, k; };
        std::vector<CS> cs;
        if (small) cs = {{C, 5 * C, 1}, {5 * C, 5 * C, 5}, {5 * C, C, 1}};
        else cs = {{C, C, 1}, {C, 5 * C, 5}, {5 * C, 5 * C, 5}, {5 * C, C, 1}};
        for (size_t r = 0; r < cs.size(); ++r) {
            int l = B.add_layer(p + P(".%d", (int)r * 3), OP_CONV, cs[r].cin, cs[r].cout, cs[r].k, true, true, true);
            int g = B.add_gn(p + P(".%d", (int)r * 3 + 1), cs[r].cout, true, true);
            add_entries_for_layer(e, l); add_entries_for_gn(e, g);
            e->decD[i].st.push_back(mk_stage(l, g, 1));
        }
    }
    {
        int l = B.add_layer("decoder.recon.0", OP_CONV, e->dec[n_st], e->N, 1, true, true, true);
        int g = B.add_gn("decoder.recon.1", e->N, true, true);
        add_entries_for_layer(e, l); add_entries_for_gn(e, g);
        e->recon.st.push_back(mk_stage(l, g, 2));
    }
    {
        e->start_lin = B.add_layer("decoder.sequence_start.0.0", OP_LINEAR, e->Z, e->Z * T, 1, true, true, false, LIN_EXPAND, e->Z);
        int l = B.add_layer("decoder.sequence_start.0.2", OP_CONV, e->Z, e->dec[0], 5, true, true, true);
        int g = B.add_gn("decoder.sequence_start.0.3", e->dec[0], true, true);
        add_entries_for_layer(e, e->start_lin); add_entries_for_layer(e, l); add_entries_for_gn(e, g);
        e->decS.st.push_back(mk_stage(l, g, 1));
    }
    for (int i = 0; i < n_st; ++i) {
        const bool live = i < n_st - 1;
        std::string p = P("decoder.xs_sequence.%d", i);
        int ll = B.add_layer(p + ".0", OP_LINEAR, e->H, e->H * T, 1, live, live, false, LIN_EXPAND, e->H);
        int l = B.add_layer(p + ".2", OP_CONV, e->H, e->dec[i + 1], 5, live, live, true);
        int g = B.add_gn(p + ".3", e->dec[i + 1], live, live);
        add_entries_for_layer(e, ll); add_entries_for_layer(e, l); add_entries_for_gn(e, g);
        e->xs_exp.push_back(ll);
        e->decX[i].st.push_back(mk_stage(l, g, 1));
    }
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < n_st; ++i) {
            const bool live = i < n_st - 1;
            const int C = (which + 1) * e->dec[i + 1];
            std::string p = P("decoder.%s.%d", which ? "condition_xz" : "condition_z", i);
            Block& b1 = which ? e->decQ1[i] : e->decP1[i];
            Block& b2 = which ? e->decQ2[i] : e->decP2[i];
            b1.residual = true;
            for (int r = 0; r < (small ? 1 : 2); ++r) {
                int l = B.add_layer(p + P(".0._seq.%d", r * 3), OP_CONV, C, C, 3, live, live, true);
                int g = B.add_gn(p + P(".0._seq.%d", r * 3 + 1), C, live, live);
                add_entries_for_layer(e, l); add_entries_for_gn(e, g);
                b1.st.push_back(mk_stage(l, g, 1));
            }
            int l2 = B.add_layer(p + ".2", OP_CONV, C, 2 * e->dec[i + 1], 3, live, live, true);
            add_entries_for_layer(e, l2);
            b2.st.push_back(mk_stage(l2, -1, 0, true, true));
        }
    }
    return 0;
}

// ---- arena layout ---------------------------------------------------------------------------
static int layout_arenas(sgv_engine* e) {
    size_t np = 0;
    auto take = [&](size_t& cur, size_t n) { size_t o = cur; cur = align_up(cur + n, 4); return o; };
    for (auto& l : e->layers) {
        l.w = take(np, (size_t)l.nw());
        l.b = take(np, l.cout);
        l.u = take(np, l.cout);
        l.v = take(np, (size_t)l.cin * l.k);
    }
    for (auto& g : e->gns) { g.gamma = take(np, g.C); g.beta = take(np, g.C); }
    e->n_params = np;
    return 0;
}

// order in which weight gradients become available during backward (for bucketed all-reduce)
static void backward_layer_order(sgv_engine* e, std::vector<std::vector<int>>& sections) {
    auto add_block = [&](std::vector<int>& v, const Block& b) {
        for (int s = (int)b.st.size() - 1; s >= 0; --s) v.push_back(b.st[s].layer);
    };
    const int n = e->n, n_st = e->n_st;
    std::vector<int> sec;
    add_block(sec, e->recon);
    sections.push_back(sec);
    for (int i = n_st - 1; i >= 0; --i) {
        sec.clear();
        if (i < n_st - 1) {
            add_block(sec, e->decQ2[i]); add_block(sec, e->decQ1[i]); add_block(sec, e->decX[i]);
            sec.push_back(e->xs_exp[i]);
            add_block(sec, e->decP2[i]); add_block(sec, e->decP1[i]);
        }
        add_block(sec, e->decD[i]); add_block(sec, e->decU[i]);
        if (i == 0) { add_block(sec, e->decS); sec.push_back(e->start_lin); }
        sections.push_back(sec);
    }
    sec.clear();
    sec.push_back(e->last_lin);
    for (int i = n - 1; i >= 1; --i) {
        if (e->layers[e->xs_lin[i]].has_grad) sec.push_back(e->xs_lin[i]);
        add_block(sec, e->encR[i]); add_block(sec, e->encA[i]);
    }
    add_block(sec, e->encR[0]);
    sections.push_back(sec);
    sec.clear();
    add_block(sec, e->encA[0]);
    sections.push_back(sec);
}

static int layout_grads(sgv_engine* e) {
    std::vector<std::vector<int>> sections;
    backward_layer_order(e, sections);
    size_t ng = 0;
    auto take = [&](size_t n) { size_t o = ng; ng = align_up(ng + n, 4); return o; };
    e->buckets.clear();
    std::vector<int> placed;          // sections that became buckets
    for (size_t si = 0; si < sections.size(); ++si) {
        size_t start = ng;
        for (int li : sections[si]) {
            Layer& l = e->layers[li];
            if (!l.has_grad) continue;
            l.gw = take((size_t)l.nw());
        }
        if (ng > start) { e->buckets.push_back({start, ng - start}); placed.push_back((int)si); }
    }
    e->n_grads_w = ng;
    size_t small_start = ng;
    // head of the small zone: the <G,W_eff> slots of the CONV layers, bucket by bucket (bucket_dots: final once the bucket's dY
    // kernels are enqueued); then the Linear layers' slots (computed from G itself at the end of backward, they travel with the
    // small bucket), biases and GroupNorm affine
    e->bucket_dots.clear();
    for (int si : placed) {
        const size_t d0 = ng;
        for (int li : sections[si]) { Layer& l = e->layers[li]; if (l.has_grad && l.op != OP_LINEAR && l.gdot == NPOS) l.gdot = take(SGV_DOT_SLOTS); }
        e->bucket_dots.push_back({d0, ng - d0});
    }
    e->dots_total = ng - small_start;
    for (auto& l : e->layers) if (l.has_grad) { if (l.gdot == NPOS) l.gdot = take(SGV_DOT_SLOTS); l.gb = take(l.cout); }
    for (auto& g : e->gns) if (g.has_grad) { g.ggamma = take(g.C); g.gbeta = take(g.C); }
    e->buckets.push_back({small_start, ng - small_start});
    e->n_grads = ng;
    // every trainable layer must have been placed
    for (auto& l : e->layers) if (l.has_grad && l.gw == NPOS) return fail(SGV_ERR_STATE, "layer %s missing from backward order", l.prefix.c_str());
    return 0;
}

// ---- activations ------------------------------------------------------------------------------
static void alloc_block(sgv_engine* e, Block& b, long M, int cin, bool need_din) {
    int c_in = cin;
    for (size_t s = 0; s < b.st.size(); ++s) {
        Stage& S = b.st[s];
        const Layer& L = e->layers[S.layer];
        if (S.pre_gelu) { S.pre = alloc_act(e, M, c_in); S.dpre = alloc_act(e, M, c_in); }
        S.y = alloc_act(e, M, L.cout, S.out_f32);
        if (S.gn >= 0 || S.act) { if (!S.a.p) S.a = alloc_act(e, M, L.cout); }
        else S.a = S.y;
        S.dy = (S.gn >= 0 || S.act) ? alloc_act(e, M, L.cout) : Tensor();
        if (s + 1 < b.st.size()) S.da = alloc_act(e, M, L.cout);
        if (S.gn >= 0) {
            const GNLayer& g = e->gns[S.gn];
            S.sums = e->n_stats_fwd; e->n_stats_fwd += (size_t)e->maxB * g.G * 2;
        }
        c_in = L.cout;
    }
    (void)need_din;
}
// Tensor.p holds arena offsets until rebase; mark "preset" views via a flag value
static void rebase(sgv_engine* e, Tensor& t) { if (t.p || t.C) t.p = e->act + (size_t)t.p; }

static int alloc_activations(sgv_engine* e) {
    const long M = (long)e->maxB * e->T;
    const int n = e->n, n_st = e->n_st;
    e->act_used = 256;   // offset 0 is reserved so that "p == 0" means unallocated
    e->x_bufs[0] = alloc_act(e, M, e->N);
    e->x_bufs[1] = alloc_act(e, M, e->N);
    e->xhat = alloc_act(e, M, e->N);
    e->dy_recon = alloc_act(e, M, e->N);
    e->enc_h.resize(n); e->d_h.resize(n);
    for (int i = 0; i < n; ++i) {
        alloc_block(e, e->encA[i], M, i == 0 ? e->N : e->enc[i - 1], i > 0);
        alloc_block(e, e->encR[i], M, e->enc[i], true);
        e->enc_h[i] = e->encR[i].st.back().a;
        e->d_h[i] = alloc_act(e, M, e->enc[i]);
    }
    e->enc_a_dummy.resize(n);
    for (int i = 0; i < n; ++i) e->enc_a_dummy[i] = alloc_act(e, M, e->enc[i]);   // d(a_i): grad wrt ConvBlock output
    e->sbuf = alloc_act(e, M, e->Z);
    e->d_sbuf = alloc_act(e, M, e->Z);
    alloc_block(e, e->decS, M, e->Z, true);
    e->zs.resize(n_st); e->dzs.resize(n_st); e->cat.resize(n_st); e->dcat.resize(n_st); e->dec_out.resize(n_st);
    e->d_out.resize(n_st); e->d_u.resize(n_st); e->d_pres.resize(n_st); e->d_qres.resize(n_st); e->d_outp.resize(n_st);
    e->gp.resize(n_st); e->gq.resize(n_st); e->xl.resize(n_st); e->d_xl.resize(n_st);
    e->zs[0] = e->decS.st.back().a;
    for (int i = 0; i < n_st; ++i) {
        const int C = e->dec[i + 1];
        const bool live = i < n_st - 1;
        if (i > 0) e->zs[i] = alloc_act(e, M, e->dec[i]);
        e->dzs[i] = alloc_act(e, M, e->dec[i]);
        alloc_block(e, e->decU[i], M, e->dec[i], true);
        if (live) {
            e->cat[i] = alloc_act(e, M, 2 * C);
            e->dcat[i] = alloc_act(e, M, 2 * C);
            // DecoderResidualBlock output and xs_sequence output are written straight into the concat buffer
            Tensor v = e->cat[i]; v.C = C; v.p = (void*)((size_t)v.p + (size_t)C * e->esz);
            e->decD[i].st.back().a = v;
            Tensor vx = e->cat[i]; vx.C = C;
            e->decX[i].st.back().a = vx;
        }
        alloc_block(e, e->decD[i], M, C, true);
        e->dec_out[i] = e->decD[i].st.back().a;
        e->d_out[i] = alloc_act(e, M, C);
        e->d_u[i] = alloc_act(e, M, C);
        if (live) {
            alloc_block(e, e->decP1[i], M, C, true);
            alloc_block(e, e->decP2[i], M, C, true);
            e->xl[i] = alloc_act(e, M, e->H);
            e->d_xl[i] = alloc_act(e, M, e->H);
            alloc_block(e, e->decX[i], M, e->H, true);
            alloc_block(e, e->decQ1[i], M, 2 * C, true);
            alloc_block(e, e->decQ2[i], M, 2 * C, true);
            e->d_pres[i] = alloc_act(e, M, C);
            e->d_qres[i] = alloc_act(e, M, 2 * C);
            e->d_outp[i] = alloc_act(e, M, C);
            e->gp[i] = alloc_act(e, M, 2 * C);
            e->gq[i] = alloc_act(e, M, 2 * C);
        }
    }
    alloc_block(e, e->recon, M, e->dec[n_st], true);
    // fp32 side buffers
    auto f32buf = [&](long count) { e->act_used = align_up(e->act_used, 256); size_t o = e->act_used; e->act_used += (size_t)count * 4; return (float*)o; };
    e->xs_raw.resize(n); e->d_xs_raw.resize(n);
    for (int i = 0; i < n; ++i) { e->xs_raw[i] = f32buf((long)e->maxB * e->H); e->d_xs_raw[i] = f32buf((long)e->maxB * e->H); }
    e->last = f32buf((long)e->maxB * 2 * e->Z); e->d_last = f32buf((long)e->maxB * 2 * e->Z);
    e->zlat = f32buf((long)e->maxB * e->Z); e->d_z = f32buf((long)e->maxB * e->Z);
    e->eps.resize(n_st); e->zmap.resize(n_st); e->eps_set.assign(n_st, 0);
    e->eps[0] = f32buf((long)e->maxB * e->Z);
    e->zmap[0] = nullptr;
    for (int i = 0; i + 1 < n_st; ++i) {
        e->eps[i + 1] = f32buf(M * e->dec[i + 1]);
        e->zmap[i] = f32buf(M * e->dec[i + 1]);
    }
    e->recon_unit = f32buf(3L * e->N);
    // backward group sums mirror the forward slots
    e->n_stats = e->n_stats_fwd * 2;
    e->act_bytes = align_up(e->act_used, 256);
    return 0;
}

static void rebase_block(sgv_engine* e, Block& b) {
    for (auto& S : b.st) {
        const bool alias = (S.gn < 0 && !S.act);
        rebase(e, S.pre); rebase(e, S.dpre); rebase(e, S.y); rebase(e, S.dy); rebase(e, S.da);
        if (alias) S.a = S.y; else rebase(e, S.a);
        if (S.sums != NPOS) S.sums2 = S.sums + e->n_stats_fwd;
    }
}
static void rebase_all(sgv_engine* e) {
    auto R = [&](Tensor& t) { rebase(e, t); };
    auto RF = [&](float*& p) { if (p) p = (float*)(e->act + (size_t)p); };
    R(e->x_bufs[0]); R(e->x_bufs[1]); e->x_cur = 0; e->x_in = e->x_bufs[0];
    R(e->xhat); R(e->dy_recon); R(e->sbuf); R(e->d_sbuf);
    for (auto& b : e->encA) rebase_block(e, b);
    for (auto& b : e->encR) rebase_block(e, b);
    for (auto& b : e->decU) rebase_block(e, b);
    for (auto& b : e->decD) rebase_block(e, b);
    for (int i = 0; i + 1 < e->n_st; ++i) { rebase_block(e, e->decP1[i]); rebase_block(e, e->decP2[i]); rebase_block(e, e->decX[i]); rebase_block(e, e->decQ1[i]); rebase_block(e, e->decQ2[i]); }
    rebase_block(e, e->decS); rebase_block(e, e->recon);
    for (auto& t : e->d_h) R(t);
    for (auto& t : e->enc_a_dummy) R(t);
    for (int i = 0; i < e->n; ++i) e->enc_h[i] = e->encR[i].st.back().a;
    for (int i = 0; i < e->n_st; ++i) {
        if (i > 0) R(e->zs[i]);
        R(e->dzs[i]); R(e->d_out[i]); R(e->d_u[i]);
        if (i + 1 < e->n_st) { R(e->cat[i]); R(e->dcat[i]); R(e->xl[i]); R(e->d_xl[i]); R(e->d_pres[i]); R(e->d_qres[i]); R(e->d_outp[i]); R(e->gp[i]); R(e->gq[i]); }
        e->dec_out[i] = e->decD[i].st.back().a;
    }
    e->zs[0] = e->decS.st.back().a;
    for (auto& p : e->xs_raw) RF(p);
    for (auto& p : e->d_xs_raw) RF(p);
    RF(e->last); RF(e->d_last); RF(e->zlat); RF(e->d_z);
    for (auto& p : e->eps) RF(p);
    for (auto& p : e->zmap) RF(p);
    RF(e->recon_unit);
}

// ---- descriptor tables --------------------------------------------------------------------------
// conv weights that train go through the tiled AdamW (optim.hip adamw_sn_kernel), which also leaves W_new^T u
// behind for the next forward's power iteration
static bool layer_fused_adam(const Layer& l) { return l.used && l.has_grad && l.op != OP_LINEAR && l.cin % 4 == 0; }
static int build_tables(sgv_engine* e) {
    // compute copies
    size_t nc = 0;
    for (auto& l : e->layers) {
        if (!l.used || l.op == OP_LINEAR) continue;
        if (e->dt == SGV_DTYPE_BF16) { l.wc = nc; nc = align_up(nc + (size_t)l.nw(), 8); }
        if (l.need_wct) { l.wct = nc; nc = align_up(nc + (size_t)l.nw(), 8); }
    }
    e->n_copies = nc;
    // SN scratch
    size_t nt = 0;
    int si = 0;
    for (auto& l : e->layers) { l.sn = si++; if (layer_fused_adam(l)) nt += align_up((size_t)l.cin * l.k, 4); }
    e->n_sn_tmp_fused = nt;
    for (auto& l : e->layers) if (!layer_fused_adam(l)) nt += align_up((size_t)l.cin * l.k, 4);
    e->sn_tmp_s_off = nt;
    for (auto& l : e->layers) nt += align_up((size_t)l.cout, 4);
    e->sn_tpart_off.clear(); e->sn_spart_off.clear();
    for (auto& l : e->layers) {
        e->sn_tpart_off.push_back(nt);
        nt += align_up((size_t)((l.cout + SN_ROWS_PER_ITEM - 1) / SN_ROWS_PER_ITEM) * l.k * l.cin, 4);
        e->sn_spart_off.push_back(nt);
        nt += align_up((size_t)l.k * ((l.cin + SN_COLS_PER_ITEM - 1) / SN_COLS_PER_ITEM) * l.cout, 4);
    }
    e->n_sn_tmp = nt;
    return 0;
}

static int upload_tables(sgv_engine* e) {
    const int L = (int)e->layers.size();
    e->sn_host.resize(L);
    size_t to_f = 0, to_u = e->n_sn_tmp_fused, to_s = e->sn_tmp_s_off;
    std::vector<WorkItem> i_sn, i_sn_unf, i_dot, i_adam, i_adam_flat, i_adam_2d, i_copy, i_wct, i_ts, i_ss;
    e->fin_lin_dots.clear();
    const int nbk = (int)e->buckets.size();
    std::vector<std::vector<WorkItem>> flat_b(nbk), tile_b(nbk), dot_b(nbk);
    std::vector<std::vector<FinDot>> fin_lin_b(nbk);
    auto bucket_of = [&](size_t goff) {
        for (int b = 0; b < nbk; ++b) if (goff >= e->buckets[b].first && goff < e->buckets[b].first + e->buckets[b].second) return b;
        return nbk - 1;
    };
    for (int i = 0; i < L; ++i) {
        Layer& l = e->layers[i];
        SNDesc d;
        d.W = e->params + l.w; d.u = e->params + l.u; d.v = e->params + l.v;
        size_t& to_t = layer_fused_adam(l) ? to_f : to_u;
        d.tmp_t = e->sn_tmp + to_t; to_t += align_up((size_t)l.cin * l.k, 4);
        d.tmp_s = e->sn_tmp + to_s; to_s += align_up((size_t)l.cout, 4);
        d.tpart = e->sn_tmp + e->sn_tpart_off[i]; d.spart = e->sn_tmp + e->sn_spart_off[i];
        d.sigma = e->sn_sigma + 2 * i;
        d.dot = l.has_grad ? e->grads + l.gdot : e->sn_dot_dummy;
        d.G = l.has_grad ? e->grads + l.gw : nullptr;
        d.wc = (e->dt == SGV_DTYPE_BF16 && l.wc != NPOS && l.cin % 8 == 0) ? (const void*)(e->copies + l.wc * e->esz) : nullptr;
        d.taps = l.k; d.rows = l.cout; d.cols = l.cin; d.active = l.used ? 1 : 0;
        e->sn_host[i] = d;
        if (l.used) {
            const int rb = (l.cout + SN_ROWS_PER_ITEM - 1) / SN_ROWS_PER_ITEM, cb = (l.cin + SN_COLS_PER_ITEM - 1) / SN_COLS_PER_ITEM;
            for (int c = 0; c < l.k * rb * cb; ++c) { i_sn.push_back({i, c}); if (!layer_fused_adam(l)) i_sn_unf.push_back({i, c}); }
            for (int c = 0; c < (l.k * l.cin + 63) / 64; ++c) i_ts.push_back({i, c});
            for (int c = 0; c < (l.cout + 63) / 64; ++c) i_ss.push_back({i, c});
        }
        if (l.has_grad && l.op == OP_LINEAR) {   // conv layers get <G,W_eff> from their dY kernels (ew.hip)
            const long nch = (l.nw() + OPT_CHUNK - 1) / OPT_CHUNK;
            const int bk = bucket_of(l.gw);
            fin_lin_b[bk].push_back({(const float*)(uintptr_t)dot_b[bk].size(), e->grads + l.gdot, (int)nch, 0});   // src = index inside the bucket for now, rebased below
            for (long c = 0; c < nch; ++c) dot_b[bk].push_back({i, (int)c});
        }
    }
    // Linear <G,W> items sorted by gradient bucket: a data-parallel backward computes a bucket's share before the bucket is
    // released (the collective may reduce the bucket's gradients in place while backward goes on)
    e->dot_off.assign(nbk + 1, 0); e->fin_lin_off.assign(nbk + 1, 0);
    for (int b = 0; b < nbk; ++b) {
        for (auto f : fin_lin_b[b]) { f.src = (const float*)((uintptr_t)f.src + i_dot.size()); e->fin_lin_dots.push_back(f); }
        i_dot.insert(i_dot.end(), dot_b[b].begin(), dot_b[b].end());
        e->dot_off[b + 1] = (int)i_dot.size(); e->fin_lin_off[b + 1] = (int)e->fin_lin_dots.size();
    }
    // the small bucket (which carries these scalars) is released before the last weight bucket: that one must hold no Linear layer
    if (nbk >= 2 && e->dot_off[nbk] != e->dot_off[nbk - 2]) return fail(SGV_ERR_STATE, "a Linear layer sits in the last weight bucket");
    e->adam_host.clear();
    auto add_adam = [&](size_t p, size_t g, long n, int sn, int rows, int cols, int taps, void* wc, void* wct, bool tiled = false) {
        AdamDesc a;
        a.p = e->params + p; a.g = e->grads + g; a.m = e->adam_m + g; a.v = e->adam_v + g;
        a.n = n; a.sn = sn; a.rows = rows; a.cols = cols; a.taps = taps; a.wc = wc; a.wct = wct;
        a.glp = nullptr;                 // option grad_bf16 points it at the bf16 mirror arena
        const int id = (int)e->adam_host.size();
        e->adam_host.push_back(a);
        const long nch = (n + OPT_CHUNK - 1) / OPT_CHUNK;
        const int bk = bucket_of(g);
        for (long c = 0; c < nch; ++c) { i_adam.push_back({id, (int)c}); if (!tiled) flat_b[bk].push_back({id, (int)c}); }
        if (tiled) {
            const int rt6 = (rows + 63) / 64, ct6 = (cols + 63) / 64;
            for (int c = 0; c < taps * rt6 * ct6; ++c) tile_b[bk].push_back({id, c});
        }
        return id;
    };
    for (int i = 0; i < L; ++i) {
        Layer& l = e->layers[i];
        void* wc = l.wc != NPOS ? (void*)(e->copies + l.wc * e->esz) : nullptr;
        void* wct = l.wct != NPOS ? (void*)(e->copies + l.wct * e->esz) : nullptr;
        int id = -1;
        if (l.has_grad) {
            id = add_adam(l.w, l.gw, l.nw(), i, l.cout, l.cin, l.k, wc, wct, layer_fused_adam(l));
            add_adam(l.b, l.gb, l.cout, -1, 1, l.cout, 1, nullptr, nullptr);
        }
        if (wc || wct) {
            if (id < 0) {   // used-in-forward but frozen layers never occur for convs; keep general
                AdamDesc a; memset(&a, 0, sizeof(a));
                a.p = e->params + l.w; a.n = l.nw(); a.sn = -1; a.rows = l.cout; a.cols = l.cin; a.taps = l.k; a.wc = wc; a.wct = wct;
                id = (int)e->adam_host.size();
                e->adam_host.push_back(a);
            }
            const int rt = (l.cout + 31) / 32, ct = (l.cin + 31) / 32;
            for (int c = 0; c < l.k * rt * ct; ++c) i_copy.push_back({id, c});
            if (wct) {
                const int rt6 = (l.cout + 63) / 64, ct6 = (l.cin + 63) / 64;
                for (int c = 0; c < l.k * rt6 * ct6; ++c) i_wct.push_back({id, c});
            }
        }
    }
    for (auto& g : e->gns) {
        if (!g.has_grad) continue;
        add_adam(g.gamma, g.ggamma, g.C, -1, 1, g.C, 1, nullptr, nullptr);
        add_adam(g.beta, g.gbeta, g.C, -1, 1, g.C, 1, nullptr, nullptr);
    }
    e->bucket_flat_w.assign(nbk, {});
    for (auto& l : e->layers) if (l.has_grad && !layer_fused_adam(l)) e->bucket_flat_w[bucket_of(l.gw)].push_back({l.gw, (size_t)l.nw()});
    e->flat_off.assign(nbk + 1, 0); e->tile_off.assign(nbk + 1, 0);
    for (int b = 0; b < nbk; ++b) {
        i_adam_flat.insert(i_adam_flat.end(), flat_b[b].begin(), flat_b[b].end());
        i_adam_2d.insert(i_adam_2d.end(), tile_b[b].begin(), tile_b[b].end());
        e->flat_off[b + 1] = (int)i_adam_flat.size(); e->tile_off[b + 1] = (int)i_adam_2d.size();
    }
    auto up = [&](const void* src, size_t bytes, void** dst) -> int {
        if (bytes == 0) { *dst = nullptr; return 0; }
        if (hipMalloc(dst, bytes) != hipSuccess) return -1;
        if (hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return -1;
        return 0;
    };
    if (up(e->sn_host.data(), sizeof(SNDesc) * L, (void**)&e->sn_dev)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(e->adam_host.data(), sizeof(AdamDesc) * e->adam_host.size(), (void**)&e->adam_dev)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_sn.data(), sizeof(WorkItem) * i_sn.size(), (void**)&e->items_sn)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_dot.data(), sizeof(WorkItem) * i_dot.size(), (void**)&e->items_dot)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_adam.data(), sizeof(WorkItem) * i_adam.size(), (void**)&e->items_adam)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_copy.data(), sizeof(WorkItem) * i_copy.size(), (void**)&e->items_copy)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_wct.data(), sizeof(WorkItem) * i_wct.size(), (void**)&e->items_wct)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_sn_unf.data(), sizeof(WorkItem) * i_sn_unf.size(), (void**)&e->items_sn_unf)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_adam_flat.data(), sizeof(WorkItem) * i_adam_flat.size(), (void**)&e->items_adam_flat)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_adam_2d.data(), sizeof(WorkItem) * i_adam_2d.size(), (void**)&e->items_adam_2d)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_ts.data(), sizeof(WorkItem) * i_ts.size(), (void**)&e->items_ts)) return fail(SGV_ERR_HIP, "table upload failed");
    if (up(i_ss.data(), sizeof(WorkItem) * i_ss.size(), (void**)&e->items_ss)) return fail(SGV_ERR_HIP, "table upload failed");
    e->n_items_ts = (int)i_ts.size(); e->n_items_ss = (int)i_ss.size();
    if (hipMalloc((void**)&e->lin_dot_part, sizeof(float) * std::max<size_t>(i_dot.size(), 1)) != hipSuccess) return fail(SGV_ERR_HIP, "hipMalloc failed");
    for (auto& f : e->fin_lin_dots) f.src = e->lin_dot_part + (size_t)(uintptr_t)f.src;
    e->n_gnorm_part = (int)std::max(i_adam_flat.size() + i_adam_2d.size(), i_adam.size());
    if (hipMalloc((void**)&e->gnorm_part, sizeof(double) * std::max(e->n_gnorm_part, 1)) != hipSuccess) return fail(SGV_ERR_HIP, "hipMalloc failed");
    if (hipMemset(e->gnorm_part, 0, sizeof(double) * std::max(e->n_gnorm_part, 1)) != hipSuccess) return fail(SGV_ERR_HIP, "memset failed");
    e->n_items_sn_unf = (int)i_sn_unf.size(); e->n_items_adam_flat = (int)i_adam_flat.size(); e->n_items_adam_2d = (int)i_adam_2d.size();
    e->n_items_sn = (int)i_sn.size(); e->n_items_dot = (int)i_dot.size();
    e->n_items_adam = (int)i_adam.size(); e->n_items_copy = (int)i_copy.size(); e->n_items_wct = (int)i_wct.size();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// timing helpers
// ------------------------------------------------------------------------------------------------
static int tag_id(sgv_engine* e, const std::string& name) {
    auto it = e->tag_ids.find(name);
    if (it != e->tag_ids.end()) return it->second;
    int id = (int)e->tag_names.size();
    e->tag_ids[name] = id; e->tag_names.push_back(name);
    return id;
}
// Times the MAIN kernel of a GEMM launch (split-K combine passes are excluded so the numbers line up with the
// rocprofv3 per-kernel averages in profiles/).
struct ScopedTimer {
    sgv_engine* e; TimerRec r; bool on; bool ended = false;
    void end_now() { if (on && !ended) { hipEventRecord(r.b, e->stream); ended = true; } }
    // detail (sgv_kernel_time_reset(e, 2)): one tag per (class, layer, shape) instead of one per class
    ScopedTimer(sgv_engine* e_, const char* cls, const Layer* l, int M = 0, int N = 0, int K = 0, int taps = 0, int sk = 0)
        : e(e_), on(e_->timing) {
        if (!on) return;
        hipEventCreate(&r.a); hipEventCreate(&r.b);
        std::string name(cls);
        if (e->timing_detail && l) {
            char buf[160];
            snprintf(buf, sizeof(buf), "|%s|M=%d N=%d K=%d taps=%d splitk=%d", l->prefix.c_str(), M, N, K, taps, sk);
            name += buf;
        }
        r.tag = tag_id(e, name);
        hipEventRecord(r.a, e->stream);
        // 128-row kernels: the launcher records the end event right after the main kernel; the 256x256 path (main kernel, its
        // split-K combine, the 128-row tail launch) is timed as a whole
        if (!strncmp(cls, "gemm_nt", 7) && strncmp(cls, "gemm_nt_t256", 12)) { gemm_nt_main_done_event(r.b); ended = true; }
    }
    ~ScopedTimer() { if (on) { end_now(); e->timers.push_back(r); } }
};

// ------------------------------------------------------------------------------------------------
// op wrappers
// ------------------------------------------------------------------------------------------------
static const void* wc_ptr(sgv_engine* e, const Layer& l) {
    if (e->dt == SGV_DTYPE_BF16) return e->copies + l.wc * e->esz;
    return e->params + l.w;
}
static const void* wct_ptr(sgv_engine* e, const Layer& l) { return e->copies + l.wct * e->esz; }

// Y = conv(X) * (1/sigma) + bias
// want_stats: let the GEMM epilogue produce the GroupNorm (sum, sum of squares) of the output per (sample, group) when the
// planned kernel can (256x256 kernel: deterministic partials + finalize; 128x128 kernel: its fp64-atomic epilogue) --
// callers check conv_fwd_fuses_stats first and skip ew_gn_stats
static void conv_fwd_params(sgv_engine* e, const Layer& l, const Tensor& x, const Tensor& y, long M, GemmNT& p) {
    memset(&p, 0, sizeof(p));
    p.A = x.p; p.lda = x.ld;
    p.W = wc_ptr(e, l); p.ldw = l.cin; p.w_tap_stride = (long)l.cout * l.cin;
    p.C = y.p; p.ldc = y.ld; p.out_f32 = y.f32 ? 1 : 0;
    p.bias = e->params + l.b;
    p.scale = e->sn_sigma + 2 * l.sn + 1;
    p.M = (int)M; p.N = l.cout; p.K = l.cin; p.taps = l.k; p.pad = (l.k - 1) / 2; p.Tlen = e->T;
    p.partial = e->partial;
}
// 0: separate statistics pass; 1: 128x128 kernel epilogue (fp64 atomics); 2: 256x256 kernel (deterministic)
static int conv_fwd_stats_mode(sgv_engine* e, const Layer& l, const Tensor& x, const Tensor& y, long M, int Cg, int G) {
    if (y.f32 || e->dt != SGV_DTYPE_BF16) return 0;
    GemmNT q; conv_fwd_params(e, l, x, y, M, q);
    q.gn_Cg = Cg; q.gn_G = G;
    const GemmPlan pl = gemm_nt_plan(e->dt, q, e->partial_floats, 1);
    if (pl.kind == 1 && pl.fuse_stats && gemm_nt256_part_floats((int)M, l.cout, 1) <= e->gn_part_floats) return 2;
    if (pl.kind == 0 && !e->deterministic && gemm_nt_can_fuse_stats(e->dt, (int)M, l.cout, l.cin, l.k, e->T, Cg)) return 1;
    return 0;
}
static bool conv_fwd_fuses_stats(sgv_engine* e, const Layer& l, const Tensor& x, const Tensor& y, long M, int Cg, int G) {
    return conv_fwd_stats_mode(e, l, x, y, M, Cg, G) != 0;
}
// A planned 256 x 256 launch whose 128-row tail would follow it as a second launch (M = 3200: rows 3072..3199): when the main
// launch leaves CUs free -- 240 work items on 256 CUs, the K = 95 008 products -- the tail runs BESIDE it instead, on the lane
// stream, as one round of <= 16 items of the 128 x 512 tile shape (each workgroup of either launch needs a CU of its own, and
// 240 + 16 = 256, so both are resident whatever the order they are placed in).  The tail is addressed by shifted row pointers
// (GemmNT::trow0 keeps the tap windows of a multi-tap product on the absolute rows); gemm_nt_tail_split decides.
static int launch_nt(sgv_engine* e, const GemmNT& p, const GemmPlan& pl) {
    // not while a collective may be resident: the tail takes exactly the CUs the main launch leaves free, and a static item list that
    // finds fewer CUs than items runs a second round
    // (kernel-timing passes take it too: the timer brackets the launch group on the main stream, join included)
    if (e->use_lanes && e->lane2 && e->tail_fork && e->stream != e->lane2 && !e->coll_inflight) {
        const int sk_t = gemm_nt_tail_split(e->dt, p, pl, e->partial_floats);
        if (sk_t > 0) {
            HIPCHK(hipEventRecord(e->tail_fork, e->stream));
            HIPCHK(hipStreamWaitEvent(e->lane2, e->tail_fork, 0));
            int r = launch_gemm_nt_main(p, pl, e->stream);
            if (r) return r;
            r = launch_gemm_nt_tail(p, pl, sk_t, e->partial2, e->lane2);
            if (r) return r;
            HIPCHK(hipEventRecord(e->tail_join, e->lane2));
            HIPCHK(hipStreamWaitEvent(e->stream, e->tail_join, 0));
            return 0;
        }
    }
    return launch_gemm_nt_planned(e->dt, p, pl, e->stream);
}
static int conv_fwd(sgv_engine* e, const Layer& l, const Tensor& x, const Tensor& y, long M, double* gn_sums = nullptr, int gn_Cg = 0,
                    int gn_G = 0) {
    GemmNT p; conv_fwd_params(e, l, x, y, M, p);
    const int smode = gn_sums ? conv_fwd_stats_mode(e, l, x, y, M, gn_Cg, gn_G) : 0;
    if (gn_sums && !smode) return fail(SGV_ERR_STATE, "conv_fwd: statistics requested from a GEMM that cannot produce them (%s)", l.prefix.c_str());
    p.gn_Cg = gn_Cg; p.gn_G = gn_G;
    GemmPlan pl = gemm_nt_plan(e->dt, p, e->partial_floats, smode == 2);
    if (smode == 2) { p.gn_sums = gn_sums; p.gn_part = e->gn_part; }
    else if (smode == 1) { p.gn_sums = gn_sums; }
    ScopedTimer tm(e, pl.kind ? "gemm_nt_t256" : gemm_nt_uses_wide(e->dt, p.N, p.K, p.taps) ? "gemm_nt_wide" : "gemm_nt", &l, p.M, p.N, p.K, p.taps, pl.sk_main);
    int r = launch_nt(e, p, pl);
    if (r) return fail(SGV_ERR_ARG, "gemm_nt launch failed for %s (M=%d N=%d K=%d, kind %d)", l.prefix.c_str(), p.M, p.N, p.K, pl.kind);
    return 0;
}
// dX = conv^T(dY) * (1/sigma) (+ addend)
static int conv_bwd_dx(sgv_engine* e, const Layer& l, const Tensor& dy, const Tensor& dx, const Tensor* addend, long M) {
    GemmNT p; memset(&p, 0, sizeof(p));
    p.A = dy.p; p.lda = dy.ld;
    p.W = wct_ptr(e, l); p.ldw = l.cout; p.w_tap_stride = (long)l.cout * l.cin;
    p.C = dx.p; p.ldc = dx.ld; p.out_f32 = 0;
    if (addend) { p.addend = addend->p; p.ldadd = addend->ld; }
    p.scale = e->sn_sigma + 2 * l.sn + 1;
    p.M = (int)M; p.N = l.cin; p.K = l.cout; p.taps = l.k; p.pad = (l.k - 1) / 2; p.Tlen = e->T;
    p.partial = e->partial;
    const GemmPlan pl = gemm_nt_plan(e->dt, p, e->partial_floats, 0);
    ScopedTimer tm(e, pl.kind ? "gemm_nt_t256" : gemm_nt_uses_wide(e->dt, p.N, p.K, p.taps) ? "gemm_nt_wide" : "gemm_nt", &l, p.M, p.N, p.K, p.taps, pl.sk_main);
    int r = launch_nt(e, p, pl);
    if (r) return fail(SGV_ERR_ARG, "gemm_nt(dX) launch failed for %s", l.prefix.c_str());
    return 0;
}
__global__ void sum_slabs_kernel(float* out, const float* partial, int splitk, long n) {
    if ((n & 3) == 0 && ((((uintptr_t)out) | ((uintptr_t)partial)) & 15) == 0) {      // 16-byte path (every conv weight)
        const long n4 = n >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            float4 v = reinterpret_cast<const float4*>(partial)[i];
            for (int z = 1; z < splitk; ++z) {
                const float4 w = reinterpret_cast<const float4*>(partial + (long)z * n)[i];
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            reinterpret_cast<float4*>(out)[i] = v;
        }
        return;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = 0.f;
        for (int z = 0; z < splitk; ++z) v += partial[(long)z * n + i];
        out[i] = v;
    }
}
static hipEvent_t next_event(sgv_engine* e) {
    if (e->ev_next == e->ev_pool.size()) {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
        e->ev_pool.push_back(ev);
    }
    return e->ev_pool[e->ev_next++];
}
// make the main stream wait for every weight-gradient GEMM issued so far on the side stream
static int join_side(sgv_engine* e) {
    if (!e->side_dirty) return 0;
    hipEvent_t ev = next_event(e);
    if (!ev) return fail(SGV_ERR_HIP, "event creation failed");
    HIPCHK(hipEventRecord(ev, e->side));
    HIPCHK(hipStreamWaitEvent(e->stream, ev, 0));
    e->side_dirty = false;
    return 0;
}
// ---- bf16 weight gradients straight from the 256 x 256 kernel (see the grad_bf16 member) ----
static bool comm_is_single(void* comm);
// which layers: those whose weight-gradient GEMM takes that kernel, unsplit, at the engine's full batch (fixed once: the AdamW table
// points the layer at the mirror arena)
static void classify_lp(sgv_engine* e) {
    if (e->lp_classified) return;
    const long M = (long)e->maxB * e->T;
    for (auto& l : e->layers) {
        l.lp = false;
        if (!(l.used && l.has_grad && l.op != OP_LINEAR && l.cin % 4 == 0) || e->dt != SGV_DTYPE_BF16 || !e->use_tr) continue;
        GemmTN q; memset(&q, 0, sizeof(q));
        q.M = (int)M; q.N1 = l.cout; q.N2 = l.cin; q.taps = l.k; q.pad = (l.k - 1) / 2; q.Tlen = e->T; q.lda = l.cout; q.ldb = l.cin; q.ldo = l.cin; q.use_tr = 1; q.splitk = 1;
        l.lp = gemm_tn_uses_t256(e->dt, q) && gemm_tn_pick_splitk(q.M, q.N1, q.N2, q.taps, e->dt, e->T) == 1;
    }
    e->bucket_lp_layers.assign(e->buckets.size(), {});
    for (size_t b = 0; b + 1 < e->buckets.size(); ++b) {
        std::vector<std::pair<size_t, int>> v;
        for (size_t i = 0; i < e->layers.size(); ++i) {
            const Layer& l = e->layers[i];
            if (l.lp && l.gw >= e->buckets[b].first && l.gw < e->buckets[b].first + e->buckets[b].second) v.push_back({l.gw, (int)i});
        }
        std::sort(v.begin(), v.end());
        for (auto& x : v) e->bucket_lp_layers[b].push_back(x.second);
    }
    e->lp_classified = true;
}
static int ensure_lp_mirror(sgv_engine* e) {
    if (!e->grads_lp) HIPCHK(hipMalloc(&e->grads_lp, e->n_grads * 2));
    classify_lp(e);
    return 0;
}
// single-GPU option: no communicator, no bucket callback
static inline bool grad_lp_active(const sgv_engine* e) { return e->grad_bf16 && e->grads_lp && !e->comm && !e->cb; }
// data-parallel step with the bf16 wire format (the condition under which fire_at packs a bucket)
static inline bool wire_lp_active(sgv_engine* e) {
    const char* off = getenv("SGV_WIRE_DIRECT");           // read per call: a test compares both forms in one process
    if (off && atoi(off) == 0) return false;
    return e->payload_bf16 && e->grads_lp && e->lp_classified && (e->comm ? !comm_is_single(e->comm) : e->cb != nullptr);
}
// the fp32 arena of the layers whose last gradient was stored as bf16: refreshed for the calls that read it.  (After a
// data-parallel step the mirror holds the averaged gradient of those layers.)
static int lp_sync(sgv_engine* e) {
    for (size_t i = 0; i < e->layers.size(); ++i) {
        if (!e->lp_dirty[i]) continue;
        const Layer& l = e->layers[i];
        ew_unpack_bf16((const char*)e->grads_lp + 2 * l.gw, e->grads + l.gw, l.nw(), e->stream);
        e->lp_dirty[i] = 0;
    }
    return 0;
}
// a bucket's fp32 gradients -> the bf16 wire copy, except the layers whose GEMM wrote the copy itself
static void pack_bucket(sgv_engine* e, int b, hipStream_t st) {
    size_t cur = e->buckets[b].first;
    const size_t end = cur + e->buckets[b].second;
    if (e->lp_classified && b < (int)e->bucket_lp_layers.size())
        for (int li : e->bucket_lp_layers[b]) {
            const Layer& l = e->layers[li];
            if (!e->lp_dirty[li]) continue;
            if (l.gw > cur) ew_pack_bf16(e->grads + cur, (char*)e->grads_lp + 2 * cur, (long)(l.gw - cur), st);
            cur = l.gw + align_up((size_t)l.nw(), 4);
        }
    if (end > cur) ew_pack_bf16(e->grads + cur, (char*)e->grads_lp + 2 * cur, (long)(end - cur), st);
}
// dW[tap][co][ci] = sum_m dY[m][co] X[m+tap-pad][ci]
static int conv_bwd_dw(sgv_engine* e, const Layer& l, const Tensor& dy, const Tensor& x, long M) {
    GemmTN p; memset(&p, 0, sizeof(p));
    p.A = dy.p; p.lda = dy.ld; p.B = x.p; p.ldb = x.ld;
    p.M = (int)M; p.N1 = l.cout; p.N2 = l.cin; p.taps = l.k; p.pad = (l.k - 1) / 2; p.Tlen = e->T;
    p.use_tr = e->use_tr;
    p.ldo = l.cin; p.out_tap_stride = (long)l.cout * l.cin;
    int sk = e->use_tr ? gemm_tn_pick_splitk(p.M, p.N1, p.N2, p.taps, e->dt, e->T) : gemm_tn_pick_splitk(p.M, p.N1, p.N2, p.taps, e->dt);
    const long nw = l.nw();
    if ((size_t)sk * nw > e->partial_tn_floats) sk = 1;
    float* G = e->grads + l.gw;
    // side stream: dY and X are final once the kernels enqueued so far on the main stream have run; nothing on the
    // main stream reads G before join_side().  (Kernel-timing passes keep everything on one stream.)
    // only the small launches go to the side stream: big GEMMs fill every CU on their own and co-running them costs L2
    static const double side_max_gf = getenv("SGV_DW_SIDE_MAXGF") ? atof(getenv("SGV_DW_SIDE_MAXGF")) : 250.0;
    const bool side = e->use_side && !e->timing && 2.0e-9 * p.M * p.N1 * p.N2 * p.taps <= side_max_gf;
    hipStream_t st = e->stream;
    float* slabs = e->partial;
    if (side) {
        hipEvent_t ev = next_event(e);
        if (!ev) return fail(SGV_ERR_HIP, "event creation failed");
        HIPCHK(hipEventRecord(ev, e->stream));
        HIPCHK(hipStreamWaitEvent(e->side, ev, 0));
        st = e->side; slabs = e->partial_tn; e->side_dirty = true;
    } else if ((size_t)sk * nw > e->partial_floats) sk = 1;
    if (e->dw_chunks > 1 && (int)(&l - e->layers.data()) == e->dw_chunk_layer && sk == 1 && !side && l.k == 1 && l.cout % (128 * e->dw_chunks) == 0) {
        const int rows = l.cout / e->dw_chunks;
        GemmTN q0 = p; q0.splitk = 1; q0.N1 = rows;
        const bool direct = l.lp && wire_lp_active(e) && gemm_tn_uses_t256(e->dt, q0);       // the chunks' wire copy straight from the GEMM
        e->dw_chunk_direct = direct;
        e->lp_dirty[(int)(&l - e->layers.data())] = direct ? 1 : 0;
        for (int c = 0; c < e->dw_chunks; ++c) {
            GemmTN q = p;
            q.splitk = 1; q.N1 = rows;
            q.A = (const char*)dy.p + (size_t)c * rows * e->esz;
            q.out = G + (size_t)c * rows * l.cin;
            if (direct) { q.out = reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(e->grads_lp) + l.gw + (size_t)c * rows * l.cin); q.out_bf16 = 1; }
            if (e->coll_inflight && e->tn_sched && !(getenv("SGV_TN256_STEAL") && atoi(getenv("SGV_TN256_STEAL")) == 0)) q.sched = e->tn_sched + 520 * (e->tn_sched_next++ & 7);
            if (launch_gemm_tn(e->dt, q, st)) return fail(SGV_ERR_ARG, "gemm_tn launch failed for %s (rows %d..%d)", l.prefix.c_str(), c * rows, (c + 1) * rows);
            if (e->dw_chunk_hook && e->dw_chunk_hook(c, e->dw_chunks, c * rows, (c + 1) * rows)) return fail(SGV_ERR_HIP, "weight-gradient chunk exchange failed for %s", l.prefix.c_str());
        }
        return 0;
    }
    ScopedTimer tm(e, "gemm_tn", &l, p.M, p.N1, p.N2, p.taps, sk);
    const int li_ = (int)(&l - e->layers.data());
    if (sk == 1) {
        p.splitk = 1; p.out = G;
        // grad_bf16: the 256 x 256 kernel rounds its accumulators to bf16 on the way out; the AdamW pass reads them there
        // a resident collective may keep some of the persistent kernel's workgroups off the chip: the work-stealing form (gemm256tn.hip)
        static const int steal_on = getenv("SGV_TN256_STEAL") ? atoi(getenv("SGV_TN256_STEAL")) : 1;
        if (steal_on && e->coll_inflight && e->tn_sched) p.sched = e->tn_sched + 520 * (e->tn_sched_next++ & 7);
        const bool opt_lp = l.lp && grad_lp_active(e), wire_lp = l.lp && !opt_lp && wire_lp_active(e);
        const bool lp = (opt_lp || wire_lp) && gemm_tn_uses_t256(e->dt, p);
        if (lp) { p.out = reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(e->grads_lp) + l.gw); p.out_bf16 = 1; }
        if (launch_gemm_tn(e->dt, p, st)) return fail(SGV_ERR_ARG, "gemm_tn launch failed for %s", l.prefix.c_str());
        // a smaller batch than the one the layer was classified at took another kernel: the optimizer still reads the mirror, so the
        // fp32 result goes there in a pass (the wire copy gets it with the rest of the bucket: pack_bucket)
        if (opt_lp && !lp) ew_pack_bf16(G, (char*)e->grads_lp + 2 * l.gw, nw, st);
        e->lp_dirty[li_] = lp ? 1 : 0;
    } else {
        // split-K over the batch*time rows: each slice writes its own fp32 slab (plain stores), then one sum pass
        p.splitk = sk; p.out = slabs; p.out_slab_stride = nw;
        if (launch_gemm_tn(e->dt, p, st)) return fail(SGV_ERR_ARG, "gemm_tn launch failed for %s", l.prefix.c_str());
        tm.end_now();
        int blocks = (int)((nw / 4 + 255) / 256); if (blocks > 4096) blocks = 4096;     // four elements per thread (grid-stride either way)
        hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, st, G, slabs, sk, nw);
        if (l.lp && grad_lp_active(e)) ew_pack_bf16(G, (char*)e->grads_lp + 2 * l.gw, nw, st);
        e->lp_dirty[li_] = 0;
    }
    return 0;
}

static GNParams gn_base(sgv_engine* e, const GNLayer& g, int B) {
    GNParams p;
    p.gamma = e->params + g.gamma; p.beta = e->params + g.beta;
    p.B = B; p.T = e->T; p.C = g.C; p.G = g.G; p.Cg = g.C / g.G;
    return p;
}

// Scope that runs the calls it encloses on the second compute lane (stream + split-K / reduction workspaces swapped in); the lane
// first waits for everything enqueued on the main stream so far.  lane2_join() makes the main stream wait for the lane.
struct Lane2 {
    sgv_engine* e; hipStream_t s0; float* p0; float* c0; float* g0; bool on;
    explicit Lane2(sgv_engine* e_) : e(e_), s0(e_->stream), p0(e_->partial), c0(e_->colpart), g0(e_->gn_part) {
        on = e->use_lanes && e->lane2 && !e->timing;
        if (!on) return;
        hipEventRecord(e->lane_fork, e->stream);
        hipStreamWaitEvent(e->lane2, e->lane_fork, 0);
        e->stream = e->lane2; e->partial = e->partial2; e->colpart = e->colpart2; e->gn_part = e->gn_part2;
    }
    ~Lane2() {
        if (!on) return;
        hipEventRecord(e->lane_join, e->lane2);
        e->stream = s0; e->partial = p0; e->colpart = c0; e->gn_part = g0;
    }
};
static void lane2_join(sgv_engine* e) {
    if (e->use_lanes && e->lane2 && !e->timing) hipStreamWaitEvent(e->stream, e->lane_join, 0);
}

static int block_fwd(sgv_engine* e, Block& b, const Tensor& in, int B) {
    const long M = (long)B * e->T;
    Tensor x = in;
    for (size_t s = 0; s < b.st.size(); ++s) {
        Stage& S = b.st[s];
        const Layer& L = e->layers[S.layer];
        Tensor cin = x;
        if (S.pre_gelu) {
            GNParams p; p.y = x.p; p.ldy = x.ld; p.out = S.pre.p; p.ldout = S.pre.ld; p.B = B; p.T = e->T; p.C = x.C;
            ew_act(e->dt, 0, p, e->stream);
            cin = S.pre;
        }
        if (S.gn >= 0 && S.act == 1 && e->use_convgn && e->dt == SGV_DTYPE_BF16 && !S.y.f32 && (long)L.cin * L.k <= e->convgn_maxk) {
            // one workgroup per (group, sample): convolution, statistics, normalise + GELU (+ residual) in one launch
            const GNLayer& g = e->gns[S.gn];
            ConvGN q; memset(&q, 0, sizeof(q));
            q.A = cin.p; q.lda = cin.ld; q.W = wc_ptr(e, L); q.ldw = L.cin; q.w_tap_stride = (long)L.cout * L.cin;
            q.bias = e->params + L.b; q.scale = e->sn_sigma + 2 * L.sn + 1;
            q.y = S.y.p; q.ldy = S.y.ld; q.out = S.a.p; q.ldout = S.a.ld;
            if (b.residual && s + 1 == b.st.size()) { q.res = in.p; q.ldres = in.ld; q.rscale = 0.1f; } else q.rscale = 1.f;
            q.gamma = e->params + g.gamma; q.beta = e->params + g.beta; q.sums = e->stats + S.sums;
            q.B = B; q.T = e->T; q.N = L.cout; q.K = L.cin; q.taps = L.k; q.pad = (L.k - 1) / 2; q.G = g.G; q.Cg = g.C / g.G;
            if (g.C == L.cout && conv_gn_fused_eligible(e->dt, q)) {
                ScopedTimer tm(e, "conv_gn", &L, (int)M, L.cout, L.cin, L.k, 1);
                if (launch_conv_gn_fwd(q, e->stream)) return fail(SGV_ERR_ARG, "conv_gn launch failed for %s", L.prefix.c_str());
                x = S.a;
                continue;
            }
        }
        CHK(conv_fwd(e, L, cin, S.y, M));
        if (S.gn >= 0) {
            const GNLayer& g = e->gns[S.gn];
            GNParams p = gn_base(e, g, B);
            p.y = S.y.p; p.ldy = S.y.ld; p.sums = e->stats + S.sums; p.part = e->colpart;
            p.out = S.a.p; p.ldout = S.a.ld;
            if (b.residual && s + 1 == b.st.size()) { p.res = in.p; p.ldres = in.ld; p.rscale = 0.1f; }
            ew_gn_fwd(e->dt, S.act, p, e->stream);
        } else if (S.act) {
            GNParams p; p.y = S.y.p; p.ldy = S.y.ld; p.out = S.a.p; p.ldout = S.a.ld; p.B = B; p.T = e->T; p.C = L.cout;
            ew_act(e->dt, 0, p, e->stream);
        }
        x = S.a;
    }
    return 0;
}

// dOut: gradient wrt the block output; dIn (nullable): gradient wrt the block input (overwritten).
// before_first_dw (optional) runs after the last dY of the block exists, right before the weight-gradient GEMM of
// the block's first conv (sgv_backward uses it to release the small-gradient bucket early).
// Input gradient of convolution L (its dY given) + GroupNorm / GELU backward of stage P below it in one launch (convgn.hip);
// `addend` (residual path) is added to the input gradient before it is rounded; `premul` = x when L reads GELU(x) (the gradient
// is multiplied by gelu'(x)); rscale = the residual scale of P's block when P is its last stage.  Returns 1 when the fused kernel ran (P.dy,
// the group sums, the per-sample column totals and the <G, W_eff> partials of P's layer are written and their fixed-order sums
// queued), 0 when the shapes are not taken, < 0 on error.
static int fused_dx_gn_bwd(sgv_engine* e, const Layer& L, const Tensor& dY, const Tensor* addend, Stage& P, int B, long M,
                           const Tensor* premul = nullptr, float rscale = 1.f, const Tensor* da_out = nullptr) {
    if (!e->use_convgn || e->dt != SGV_DTYPE_BF16 || !L.need_wct || (long)L.cout * L.k > e->convgn_maxk) return 0;
    if (P.gn < 0 || P.act != 1 || P.y.f32) return 0;
    const Layer& LP = e->layers[P.layer];
    const GNLayer& g = e->gns[P.gn];
    ConvGNBwd q; memset(&q, 0, sizeof(q));
    q.A = dY.p; q.lda = dY.ld; q.W = wct_ptr(e, L); q.ldw = L.cout; q.w_tap_stride = (long)L.cin * L.cout;
    q.scale = e->sn_sigma + 2 * L.sn + 1;
    if (addend) { q.addend = addend->p; q.ldadd = addend->ld; }
    if (premul) { q.premul = premul->p; q.ldpre = premul->ld; }
    if (da_out) { q.da = da_out->p; q.ldda = da_out->ld; }
    q.y = P.y.p; q.ldy = P.y.ld; q.sums = e->stats + P.sums; q.gamma = e->params + g.gamma; q.beta = e->params + g.beta;
    q.cbias = e->params + LP.b; q.dy = P.dy.p; q.lddy = P.dy.ld; q.sums2 = e->stats + P.sums2; q.ptot = e->red + g.ptot;
    q.cdot_part = e->red + LP.dot_part; q.rscale = rscale; q.gscale = 1.f;
    q.B = B; q.T = e->T; q.N = L.cin; q.K = L.cout; q.taps = L.k; q.pad = (L.k - 1) / 2; q.G = g.G; q.Cg = g.C / g.G;
    if (g.C != L.cin || LP.cout != L.cin || !conv_gn_bwd_eligible(e->dt, q)) return 0;
    ScopedTimer tm(e, "conv_gn_bwd", &L, (int)M, L.cin, L.cout, L.k, 1);
    if (launch_conv_gn_bwd(q, e->stream)) return fail(SGV_ERR_ARG, "conv_gn_bwd launch failed for %s", L.prefix.c_str());
    int* cntp = &e->dot_counts[e->fin_dots.size() % 512];
    *cntp = g.G * B;
    e->fin_dots.push_back({q.cdot_part, e->grads + LP.gdot, *cntp, 0});
    e->fin_affine.push_back({q.ptot, e->grads + g.gbeta, e->grads + g.ggamma, e->grads + LP.gb, g.C, B, 0, 0});
    return 1;
}

// below / below_done (optional): the last stage of the block that consumes dIn as its incoming gradient; when the fused kernel
// can take (this block's first convolution, that stage) together, it writes dIn AND that stage's dY, *below_done is set and the
// caller passes last_dy_ready = true to that block's block_bwd (below_rscale: that block's residual scale).
static int block_bwd(sgv_engine* e, Block& b, const Tensor& in, const Tensor& dOut, const Tensor* dIn, int B,
                     const std::function<void()>* before_first_dw = nullptr, Stage* below = nullptr, bool* below_done = nullptr,
                     bool last_dy_ready = false, float below_rscale = 1.f) {
    const long M = (long)B * e->T;
    Tensor dA = dOut;
    float sc = b.residual ? 0.1f : 1.0f;
    bool dy_ready = last_dy_ready;  // the stage's dY was produced by the fused kernel launched from the stage above
    for (int s = (int)b.st.size() - 1; s >= 0; --s) {
        Stage& S = b.st[s];
        const Layer& L = e->layers[S.layer];
        const Tensor x_raw = (s == 0) ? in : b.st[s - 1].a;
        const Tensor x_conv = S.pre_gelu ? S.pre : x_raw;
        Tensor dY;
        // <G,W_eff> and the GroupNorm affine / bias gradients leave these kernels as block / per-sample partials in e->red; the
        // fixed-order sums run once per bucket (flush_fin in backward_impl)
        int* cnt = &e->dot_counts[e->fin_dots.size() % 512];
        if (e->recompute_act && S.gn >= 0 && !S.y.f32) {
            // regenerate this stage's output map exactly as block_fwd's unfused path writes it (the fused forward kernel normalises the
            // same stored values): the map is read below as the next stage's convolution input (weight gradient) and by the
            // residual adds; a recompute build would not have kept it
            const GNLayer& g = e->gns[S.gn];
            GNParams p = gn_base(e, g, B);
            p.y = S.y.p; p.ldy = S.y.ld; p.sums = e->stats + S.sums; p.part = e->colpart;
            p.out = S.a.p; p.ldout = S.a.ld;
            if (b.residual && s + 1 == (int)b.st.size()) { p.res = in.p; p.ldres = in.ld; p.rscale = 0.1f; }
            if (ew_gn_apply(e->dt, S.act, p, e->stream)) return fail(SGV_ERR_HIP, "activation recompute launch failed (%s)", L.prefix.c_str());
            e->recompute_bytes += (size_t)M * g.C * e->esz;
        }
        if (dy_ready) {
            dY = S.dy;
            dy_ready = false;
        } else if (S.gn >= 0) {
            const GNLayer& g = e->gns[S.gn];
            GNParams p = gn_base(e, g, B);
            p.y = S.y.p; p.ldy = S.y.ld; p.sums = e->stats + S.sums; p.sums2 = e->stats + S.sums2;
            p.dout = dA.p; p.lddout = dA.ld; p.rscale = sc;
            p.part = e->colpart;
            p.ptot = e->red + g.ptot;
            p.out = S.dy.p; p.ldout = S.dy.ld;
            p.cdot = e->grads + L.gdot; p.cbias = e->params + L.b;   // <G,W_eff> = sum dY*(y - bias)
            p.cdot_part = e->red + L.dot_part; p.cdot_blocks = cnt;
            if (ew_gn_bwd(e->dt, 1, p, e->stream)) return fail(SGV_ERR_STATE, "GroupNorm backward launch failed (%s)", L.prefix.c_str());           // sums2, dY (GELU), partials
            e->fin_dots.push_back({p.cdot_part, p.cdot, *cnt, 0});
            e->fin_affine.push_back({p.ptot, e->grads + g.gbeta, e->grads + g.ggamma, e->grads + L.gb, g.C, B, 0, 0});
            dY = S.dy;
        } else if (S.act) {
            GNParams p; p.y = S.y.p; p.ldy = S.y.ld; p.dout = dA.p; p.lddout = dA.ld; p.rscale = sc;
            p.out = S.dy.p; p.ldout = S.dy.ld; p.dbias = e->grads + L.gb; p.part = e->colpart; p.B = B; p.T = e->T; p.C = L.cout;
            p.cdot = e->grads + L.gdot; p.cbias = e->params + L.b;
            p.cdot_part = e->red + L.dot_part; p.cdot_blocks = cnt;
            if (L.col_part != NPOS) { p.part = e->red + L.col_part; p.defer_colsum = 1; }     // bias gradient: summed with the small bucket's other sums
            ew_act(e->dt, 1, p, e->stream);
            if (L.col_part != NPOS) e->fin_affine.push_back({p.part, p.dbias, nullptr, nullptr, L.cout, ew_act_part_rows(B, e->T, L.cout), 0, 1});
            e->fin_dots.push_back({p.cdot_part, p.cdot, *cnt, 0});
            dY = S.dy;
        } else {
            GNParams p; p.y = dA.p; p.ldy = dA.ld; p.dbias = e->grads + L.gb; p.part = e->colpart; p.B = B; p.T = e->T; p.C = L.cout;
            if (!S.y.f32) return fail(SGV_ERR_STATE, "conv without norm/activation must have an fp32 output (%s)", L.prefix.c_str());
            p.cdot = e->grads + L.gdot; p.cbias = e->params + L.b; p.yf32 = (const float*)S.y.p; p.ldyf = S.y.ld;
            p.cdot_part = e->red + L.dot_part; p.cdot_blocks = cnt;
            if (L.col_part != NPOS) { p.part = e->red + L.col_part; p.defer_colsum = 1; }
            ew_act(e->dt, 2, p, e->stream);
            if (L.col_part != NPOS) e->fin_affine.push_back({p.part, p.dbias, nullptr, nullptr, L.cout, ew_act_part_rows(B, e->T, L.cout), 0, 1});
            e->fin_dots.push_back({p.cdot_part, p.cdot, *cnt, 0});
            dY = dA;
        }
        sc = 1.0f;
        if (s == 0 && before_first_dw) (*before_first_dw)();
        CHK(conv_bwd_dw(e, L, dY, x_conv, M));
        const bool need = (s > 0) || (dIn != nullptr);
        if (need && (s > 0 ? !S.pre_gelu : (below && below_done))) {
            Stage& P = s > 0 ? b.st[s - 1] : *below;
            const Tensor* add = (s == 0 && b.residual) ? &dOut : nullptr;
            // across blocks the input gradient itself is stored too (dIn): a residual block below adds it to its own input gradient
            const int fr = fused_dx_gn_bwd(e, L, dY, add, P, B, M, S.pre_gelu ? &x_raw : nullptr, s > 0 ? 1.f : below_rscale, s > 0 ? nullptr : dIn);
            if (fr < 0) return fr;
            if (fr > 0) {
                if (s > 0) { dy_ready = true; dA = P.da; }
                else *below_done = true;
                continue;
            }
        }
        if (need) {
            const Tensor target = (s > 0) ? b.st[s - 1].da : *dIn;
            if (S.pre_gelu) {
                CHK(conv_bwd_dx(e, L, dY, S.dpre, nullptr, M));
                GNParams p; p.y = x_raw.p; p.ldy = x_raw.ld; p.dout = S.dpre.p; p.lddout = S.dpre.ld; p.rscale = 1.f;
                p.out = target.p; p.ldout = target.ld; p.B = B; p.T = e->T; p.C = x_raw.C;
                ew_act(e->dt, 1, p, e->stream);
            } else {
                const Tensor* add = (s == 0 && b.residual) ? &dOut : nullptr;
                CHK(conv_bwd_dx(e, L, dY, target, add, M));
            }
            dA = target;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* sgv_last_error(void) { return g_err; }

// Auxiliary streams.  The HIP runtime maps streams onto a handful of hardware queues PER PRIORITY LEVEL (GPU_MAX_HW_QUEUES = 4),
// round-robin in creation order, and two streams on one queue run their kernels strictly one after the other: a kernel trace showed
// the second compute lane and the collective's stream sharing the main stream's queue (no overlap at all) depending on how many
// streams the process had created before.  A stream of another priority level comes from another queue pool, so it can never land
// on the main stream's queue: level -1 = high, 0 = normal (the main stream's), 1 = low; SGV_PRIO_* override the defaults.
constexpr int SGV_PRIO_LANE_DEFAULT = 0, SGV_PRIO_SIDE_DEFAULT = 0, SGV_PRIO_OPT_DEFAULT = 0;
static hipError_t make_stream(hipStream_t* s, const char* env, int level);
// ---- which hardware queue did a new stream land on? ----
// Not visible through the API, but observable: a kernel on stream b cannot finish while a kernel on stream a spins if both sit
// on one queue.  probe_spin_kernel waits on the constant-rate clock for a bounded time (always exits), probe_nop_kernel is empty.
__global__ void probe_spin_kernel(long long ticks) {
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ void probe_nop_kernel() {}
// One probe: 1 = kernels of a and b run concurrently (different hardware queues), 0 = b's kernel finished only after a's, -1 = API
// error.  Decided by the ORDER of two device-side timestamps (the event behind the spin on a, the event behind the empty kernel
// on b), not by host wall time: if b's kernel ended while a was still spinning, the queues are different.
static int streams_overlap_once(hipStream_t a, hipStream_t b, long long ticks) {
    hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) return -1;
    if (hipEventCreate(&ea) != hipSuccess) { hipEventDestroy(e0); return -1; }
    if (hipEventCreate(&eb) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(ea); return -1; }
    int res = -1;
    if (hipEventRecord(e0, a) == hipSuccess) {
        hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, a, ticks);
        if (hipEventRecord(ea, a) == hipSuccess) {
            hipLaunchKernelGGL(probe_nop_kernel, dim3(1), dim3(64), 0, b);
            float ta = 0.f, tb = 0.f;       // both measured from e0, which precedes both kernels: never a negative interval
            if (hipEventRecord(eb, b) == hipSuccess && hipStreamSynchronize(b) == hipSuccess && hipStreamSynchronize(a) == hipSuccess &&
                hipEventElapsedTime(&ta, e0, ea) == hipSuccess && hipEventElapsedTime(&tb, e0, eb) == hipSuccess)
                res = tb < ta - 0.02f ? 1 : 0;        // b's kernel was over >= 20 us before the spin ended
        }
    }
    hipStreamSynchronize(a);
    hipEventDestroy(e0); hipEventDestroy(ea); hipEventDestroy(eb);
    (void)hipGetLastError();
    return res;
}
// true: kernels of a and b run concurrently; on any API error: true (no reason to reject the stream).  The candidate gets an untimed
// first launch (a new stream's first launch can take longer than the spin), and a "shares a queue" verdict is confirmed once with a
// ten times longer spin: a host that needed more than 300 us to submit the empty kernel (loaded box, profiler attached) would
// otherwise reject a good candidate.
static bool streams_overlap(hipStream_t a, hipStream_t b) {
    hipLaunchKernelGGL(probe_nop_kernel, dim3(1), dim3(64), 0, b);
    if (hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return true; }
    int r = streams_overlap_once(a, b, 30000LL);              // 300 us at the 100 MHz constant clock
    if (r == 0) r = streams_overlap_once(a, b, 300000LL);     // 3 ms
    return r != 0;
}
// A new auxiliary stream that shares its hardware queue with none of `avoid`.  The runtime gives a new stream the least-loaded
// queue (round-robin in a fresh process; in a process that has created and destroyed many streams the main stream's queue can be the
// emptiest for many creations in a row), so the rejected candidates stay alive until a keeper is found -- every reject loads the
// queue it sits on and steers the next candidate elsewhere -- and up to 32 candidates are tried (0.3 ms each per stream to avoid).
// SGV_STREAM_PROBE=0: take the first.  If every candidate collides the last one is kept.
static hipError_t make_aux_stream(hipStream_t* out, const char* env, int level, std::initializer_list<hipStream_t> avoid) {
    static const int probe = getenv("SGV_STREAM_PROBE") ? atoi(getenv("SGV_STREAM_PROBE")) : 1;
    std::vector<hipStream_t> rejected;
    hipStream_t s = nullptr;
    hipError_t rc = hipSuccess;
    constexpr int kAttempts = 32;
    for (int attempt = 0; attempt < kAttempts; ++attempt) {
        s = nullptr;
        rc = make_stream(&s, env, level);
        if (rc != hipSuccess || !probe) break;
        bool ok = true;
        for (hipStream_t a : avoid) if (a != s && !streams_overlap(a, s)) { ok = false; break; }      // a == nullptr is the null stream: probed too
        if (getenv("SGV_STREAM_LOG")) fprintf(stderr, "[sgvae] %s: candidate %d %s\n", env, attempt, ok ? "kept" : "shares a hardware queue with a stream it must not, rejected");
        if (ok) break;
        if (attempt == kAttempts - 1) {
            // kept all the same: the engine stays correct, but this stream's kernels now run between the other stream's instead of
            // beside them (no lane / optimizer / communication overlap) -- say so once, the bench line reports it as well
            fprintf(stderr, "[sgvae] warning: %s: all %d candidate streams share a hardware queue with a stream they must avoid "
                            "(GPU_MAX_HW_QUEUES too small for this process?); overlap on this stream is lost\n", env, kAttempts);
            break;
        }
        rejected.push_back(s);
    }
    for (hipStream_t r : rejected) hipStreamDestroy(r);
    *out = rc == hipSuccess ? s : nullptr;
    return rc;
}
static hipStream_t ensure_opt(sgv_engine* e) {
    // never on the main stream's queue: an AdamW launch that waits for a collective there would hold back every kernel behind it
    if (!e->opt && make_aux_stream(&e->opt, "SGV_PRIO_OPT", SGV_PRIO_OPT_DEFAULT, {e->stream, e->side, e->lane2}) != hipSuccess) e->opt = nullptr;
    return e->opt;
}
// a communication stream for sgv_set_rccl that is guaranteed not to sit on the main stream's hardware queue (a collective there
// would run strictly between the main stream's kernels instead of beside them)
static hipStream_t ensure_comm_own(sgv_engine* e) {
    if (!e->comm_own && make_aux_stream(&e->comm_own, "SGV_PRIO_COMM", 0, {e->stream, e->side, e->lane2}) != hipSuccess) e->comm_own = nullptr;
    return e->comm_own;
}
static hipStream_t ensure_wire(sgv_engine* e) {
    if (!e->wire && make_aux_stream(&e->wire, "SGV_PRIO_WIRE", SGV_PRIO_OPT_DEFAULT, {e->stream, e->side}) != hipSuccess) e->wire = nullptr;
    return e->wire;
}
// ---- prefetched augmentation (see the members) ----
static constexpr size_t AUG_CTL = 8192;
static bool ensure_aug(sgv_engine* e) {
    if (e->aug_stream) return true;
    if (make_aux_stream(&e->aug_stream, "SGV_PRIO_AUG", 0, {e->stream, e->side, e->lane2}) != hipSuccess) { e->aug_stream = nullptr; return false; }
    bool ok = hipEventCreateWithFlags(&e->aug_done, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&e->aug_gate, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) ok = hipEventCreateWithFlags(&e->x_free[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&e->aug_ctl, AUG_CTL) == hipSuccess;
    if (!ok) { hipStreamDestroy(e->aug_stream); e->aug_stream = nullptr; }
    return ok;
}
// Enqueue the staged batch's kernels, ordered after the current position of the stream the caller enqueues on.  A few samples per
// launch: one launch of 8192 small workgroups would refill every CU as slots free and keep the forward pass's big-LDS workgroups
// (fused Conv+GroupNorm stages, 128-row GEMMs) off the chip until it ends (AdamW slices).
static int aug_fire(sgv_engine* e) {
    if (!e->aug_staged || e->aug_fired) return SGV_OK;
    static const int per = getenv("SGV_AUG_SLICE") ? std::max(1, atoi(getenv("SGV_AUG_SLICE"))) : 2;
    const int nb = 1 - e->x_cur, batch = e->aug_next_batch;
    char* scratch = e->aug_ctl;
    int* d_idx = (int*)scratch; int* d_mix = d_idx + batch;
    float* d_scale = (float*)(d_mix + batch); float* d_lam = d_scale + batch;
    unsigned long long* d_seed = (unsigned long long*)(scratch + align_up((size_t)batch * 16, 8));
    HIPCHK(hipEventRecord(e->aug_gate, e->stream));
    HIPCHK(hipStreamWaitEvent(e->aug_stream, e->aug_gate, 0));
    const long se = (long)e->N * e->T;
    for (int b0 = 0; b0 < batch; b0 += per) {
        const int nbt = std::min(per, batch - b0);
        ew_augment(e->dt, e->aug_data, (char*)e->x_bufs[nb].p + (size_t)b0 * se * e->esz, se, nbt, d_idx + b0, d_seed + b0, d_scale + b0, d_mix + b0, d_lam + b0, e->aug_stream);
    }
    HIPCHK(hipEventRecord(e->aug_done, e->aug_stream));
    if (e->timing) HIPCHK(hipStreamWaitEvent(e->stream, e->aug_done, 0));      // kernel-timing passes keep the step on one stream
    e->aug_fired = true;
    return SGV_OK;
}
// the main stream takes the batch whose prefetch kernels may still be running
static int aug_join(sgv_engine* e) {
    if (e->aug_pending) { HIPCHK(hipStreamWaitEvent(e->stream, e->aug_done, 0)); e->aug_pending = false; }
    return SGV_OK;
}
// the main stream is done reading the current input buffer (end of a forward pass, end of backward)
static void x_release(sgv_engine* e) {
    if (!e->aug_stream) return;
    if (hipEventRecord(e->x_free[e->x_cur], e->stream) == hipSuccess) e->x_free_set[e->x_cur] = true;
}
static hipError_t make_stream(hipStream_t* s, const char* env, int level) {
    if (getenv(env)) level = atoi(getenv(env));
    if (level == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, level < 0 ? greatest : least);
}
int sgv_create(const sgv_config* cfg, void* hip_stream, sgv_engine** out) {
    if (!cfg || !out) return fail(SGV_ERR_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SGV_ERR_NOGPU, "no HIP device visible: libsgvae has no CPU fallback");
    if (cfg->n_levels < 2 || cfg->n_levels > SGV_MAX_LEVELS) return fail(SGV_ERR_ARG, "n_levels must be in [2,%d]", SGV_MAX_LEVELS);
    if (cfg->n_levels - 1 + 2 > SGV_MAX_SCALARS) return fail(SGV_ERR_ARG, "too many levels");
    if (cfg->compute_dtype != SGV_DTYPE_F32 && cfg->compute_dtype != SGV_DTYPE_BF16) return fail(SGV_ERR_ARG, "bad compute_dtype");
    if (cfg->num_node % 8 || cfg->latent_dim % 8 || cfg->hierarchical_dim % 8)
        return fail(SGV_ERR_ARG, "num_node, latent_dim and hierarchical_dim must be multiples of 8 (16-byte channel vectors)");
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->num_filter_enc[i] % 8 || cfg->num_filter_enc[i] <= 0) return fail(SGV_ERR_ARG, "num_filter_enc[%d]=%d must be a positive multiple of 8", i, cfg->num_filter_enc[i]);
    if (cfg->max_batch < 1 || cfg->num_time < 1) return fail(SGV_ERR_ARG, "bad batch/time");
    if (cfg->loss_type < 0 || cfg->loss_type > 3) return fail(SGV_ERR_ARG, "bad loss_type");
    sgv_engine* e = new sgv_engine();
    e->cfg = *cfg;
    e->stream = (hipStream_t)hip_stream;
    e->dt = cfg->compute_dtype; e->esz = e->dt == SGV_DTYPE_BF16 ? 2 : 4;
    e->n = cfg->n_levels; e->n_st = e->n - 1; e->T = cfg->num_time; e->N = cfg->num_node;
    e->Z = cfg->latent_dim; e->H = cfg->hierarchical_dim; e->maxB = cfg->max_batch;
    e->use_tr = (cfg->flags & 1) ? 0 : 1;
    for (int i = 0; i < e->n; ++i) e->enc.push_back(cfg->num_filter_enc[i]);
    e->dec.assign(e->enc.rbegin(), e->enc.rend());
    int r;
    if ((r = build_graph(e)) || (r = layout_arenas(e)) || (r = layout_grads(e)) || (r = alloc_activations(e)) || (r = build_tables(e))) { delete e; return r; }
    // workspace for split-K slabs: enough for the largest split GEMM
    const long M = (long)e->maxB * e->T;
    size_t pf = 0, pf_tn = 0;
    for (auto& l : e->layers) {
        if (!l.used || l.op == OP_LINEAR) continue;
        int sk = gemm_nt_pick_splitk((int)M, l.cout, l.cin, l.k, e->dt);
        if (sk > 1) pf = std::max(pf, (size_t)sk * M * l.cout);
        sk = gemm_nt_pick_splitk((int)M, l.cin, l.cout, l.k, e->dt);
        if (sk > 1 && l.need_wct) pf = std::max(pf, (size_t)sk * M * l.cin);
        sk = std::max(gemm_tn_pick_splitk((int)M, l.cout, l.cin, l.k, e->dt, e->T), gemm_tn_pick_splitk((int)M, l.cout, l.cin, l.k, e->dt));
        if (sk > 1) pf_tn = std::max(pf_tn, (size_t)sk * l.nw());
    }
    if (pf < ((size_t)32 << 20)) pf = (size_t)32 << 20;   // batch < max_batch can pick deeper splits
    if (pf_tn < ((size_t)32 << 20)) pf_tn = (size_t)32 << 20;
    e->partial_floats = pf;
    e->partial_tn_floats = pf_tn;
    e->colpart_floats = 0;
    for (auto& g : e->gns) e->colpart_floats = std::max(e->colpart_floats, ew_gn_part_floats(e->maxB, e->T, g.C));
    for (auto& l : e->layers) if (l.op != OP_LINEAR) e->colpart_floats = std::max(e->colpart_floats, ew_gn_part_floats(e->maxB, e->T, l.cout));
    e->xpose_floats = (size_t)M * std::max(e->N, 8);
    for (int i = 0; i < e->n; ++i) e->xpose_floats = std::max(e->xpose_floats, (size_t)M * e->enc[i] * 2);
#define ALLOC(ptr, bytes)                                                                                      \
    do {                                                                                                       \
        size_t b_ = (bytes);                                                                                   \
        if (b_ == 0) b_ = 256;                                                                                 \
        if (hipMalloc((void**)&(ptr), b_) != hipSuccess) { int rc = fail(SGV_ERR_HIP, "hipMalloc(%zu bytes) failed for " #ptr, b_); sgv_destroy(e); return rc; } \
        hipMemsetAsync((ptr), 0, b_, e->stream);                                                               \
    } while (0)
    ALLOC(e->params, e->n_params * 4);
    ALLOC(e->grads, e->n_grads * 4);
    e->lp_dirty.assign(e->layers.size(), 0);
    ALLOC(e->adam_m, e->n_grads * 4);
    ALLOC(e->adam_v, e->n_grads * 4);
    ALLOC(e->copies, e->n_copies * e->esz);
    ALLOC(e->act, e->act_bytes);
    ALLOC(e->stats, e->n_stats * 8);
    ALLOC(e->sn_tmp, e->n_sn_tmp * 4);
    ALLOC(e->sn_sigma, e->layers.size() * 2 * 4);
    ALLOC(e->sn_dot_dummy, SGV_DOT_SLOTS * sizeof(float));
    ALLOC(e->scal, 32 * 8);
    ALLOC(e->partial, e->partial_floats * 4);
    ALLOC(e->partial_tn, e->partial_tn_floats * 4);
    e->gn_part_floats = 0;
    for (auto& l : e->layers) if (l.used && l.op != OP_LINEAR) e->gn_part_floats = std::max(e->gn_part_floats, gemm_nt256_part_floats((int)M, l.cout, 1));
    ALLOC(e->gn_part, e->gn_part_floats * 4);
    if (e->use_lanes && make_aux_stream(&e->lane2, "SGV_PRIO_LANE", SGV_PRIO_LANE_DEFAULT, {e->stream}) == hipSuccess &&
        hipEventCreateWithFlags(&e->lane_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&e->lane_join, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&e->tail_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&e->tail_join, hipEventDisableTiming) == hipSuccess) {
        ALLOC(e->partial2, e->partial_floats * 4);
        ALLOC(e->gn_part2, e->gn_part_floats * 4);
    } else {
        e->use_lanes = 0;
    }
    {
        size_t nr = 0;
        for (auto& g : e->gns) { g.ptot = nr; nr += align_up((size_t)e->maxB * 3 * g.C, 4); }
        for (auto& l : e->layers) if (l.op != OP_LINEAR) { l.dot_part = nr; nr += align_up((size_t)ew_gn_max_blocks(e->maxB, e->T, l.cout), 4); }
        for (auto& l : e->layers) if (l.op != OP_LINEAR && l.used && l.cout <= 8192) { l.col_part = nr; nr += align_up((size_t)ew_act_part_rows(e->maxB, e->T, l.cout) * l.cout, 4); }
        e->red_floats = nr;
    }
    ALLOC(e->red, e->red_floats * 4);
    if (make_aux_stream(&e->side, "SGV_PRIO_SIDE", SGV_PRIO_SIDE_DEFAULT, {e->stream, e->lane2}) != hipSuccess) { e->side = nullptr; e->use_side = false; }
    // the optimizer and wire streams of the data-parallel step are created on first use (ensure_opt / ensure_wire): every stream a
    // process creates shifts the runtime's stream -> hardware-queue assignment of the ones created after it
    if (getenv("SGV_DW_SIDE")) e->use_side = atoi(getenv("SGV_DW_SIDE")) != 0 && e->side != nullptr;
    ALLOC(e->xpose_tmp, e->xpose_floats * 4);
    ALLOC(e->colpart, e->colpart_floats * 4);
    if (e->use_lanes) ALLOC(e->colpart2, e->colpart_floats * 4);
#undef ALLOC
    rebase_all(e);
    if ((r = upload_tables(e))) { sgv_destroy(e); return r; }
    if (hipStreamSynchronize(e->stream) != hipSuccess) { sgv_destroy(e); return fail(SGV_ERR_HIP, "stream sync failed in create"); }
    *out = e;
    return SGV_OK;
}

int sgv_destroy(sgv_engine* e) {
    if (!e) return SGV_OK;
    hipStreamSynchronize(e->stream);
    void* ptrs[] = {e->params, e->grads, e->adam_m, e->adam_v, e->copies, e->act, e->stats, e->sn_tmp, e->sn_sigma, e->sn_dot_dummy,
                    e->scal, e->partial, e->partial_tn, e->partial2, e->colpart2, e->gn_part2, e->gn_part, e->red, e->xpose_tmp, e->colpart, e->sn_dev, e->adam_dev, e->items_sn, e->items_dot, e->items_adam, e->items_copy, e->items_wct,
                    e->items_sn_unf, e->items_adam_flat, e->items_adam_2d, e->items_ts, e->items_ss, e->lin_dot_part, e->gnorm_part};
    for (void* p : ptrs) if (p) hipFree(p);
    if (e->side) { hipStreamSynchronize(e->side); hipStreamDestroy(e->side); }
    if (e->opt) { hipStreamSynchronize(e->opt); hipStreamDestroy(e->opt); }
    if (e->wire) { hipStreamSynchronize(e->wire); hipStreamDestroy(e->wire); }
    if (e->comm_own) { hipStreamSynchronize(e->comm_own); hipStreamDestroy(e->comm_own); }
    if (e->lane2) { hipStreamSynchronize(e->lane2); hipStreamDestroy(e->lane2); }
    if (e->aug_stream) {
        hipStreamSynchronize(e->aug_stream); hipStreamDestroy(e->aug_stream);
        hipEventDestroy(e->aug_done); hipEventDestroy(e->aug_gate); hipEventDestroy(e->x_free[0]); hipEventDestroy(e->x_free[1]); hipFree(e->aug_ctl);
    }
    if (e->lane_fork) hipEventDestroy(e->lane_fork);
    if (e->tail_fork) hipEventDestroy(e->tail_fork);
    if (e->tail_join) hipEventDestroy(e->tail_join);
    if (e->lane_join) hipEventDestroy(e->lane_join);
    for (auto ev : e->ev_pool) hipEventDestroy(ev);
    for (auto& t : e->timers) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    if (e->grads_lp) hipFree(e->grads_lp);
    if (e->tn_sched) hipFree(e->tn_sched);
    delete e;
    return SGV_OK;
}

int sgv_param_count(const sgv_engine* e) { return e ? (int)e->entries.size() : 0; }

int sgv_param_info(const sgv_engine* e, int index, const char** name, int* ndim, int64_t shape[4], int* kind, int* has_grad) {
    if (!e || index < 0 || index >= (int)e->entries.size()) return fail(SGV_ERR_ARG, "bad index");
    const StateEntry& s = e->entries[index];
    if (name) *name = s.name.c_str();
    if (ndim) *ndim = (int)s.shape.size();
    if (shape) for (size_t i = 0; i < s.shape.size() && i < 4; ++i) shape[i] = s.shape[i];
    if (kind) *kind = s.kind;
    if (has_grad) *has_grad = s.has_grad ? 1 : 0;
    return SGV_OK;
}

}  // extern "C"

// ---- reference layout <-> internal layout (host) ---------------------------------------------
// dir = +1: ref -> internal; -1: internal -> ref
static void permute_entry(const sgv_engine* e, const StateEntry& s, const float* src, float* dst, int dir) {
    const long cnt = s.count();
    if (s.gn >= 0) { memcpy(dst, src, cnt * 4); return; }
    const Layer& l = e->layers[s.layer];
    const int T = e->T;
    auto mov = [&](long iref, long iint) { if (dir > 0) dst[iint] = src[iref]; else dst[iref] = src[iint]; };
    if (s.kind == 1) {
        if (l.op == OP_CONV) {
            for (int co = 0; co < l.cout; ++co) for (int ci = 0; ci < l.cin; ++ci) for (int j = 0; j < l.k; ++j)
                mov(((long)co * l.cin + ci) * l.k + j, ((long)j * l.cout + co) * l.cin + ci);
        } else if (l.op == OP_CONVT) {
            for (int ci = 0; ci < l.cin; ++ci) for (int co = 0; co < l.cout; ++co) for (int j = 0; j < l.k; ++j)
                mov(((long)ci * l.cout + co) * l.k + j, ((long)(l.k - 1 - j) * l.cout + co) * l.cin + ci);
        } else if (l.lin_kind == LIN_HEAD) {
            const int C = l.lin_C;
            for (int o = 0; o < l.cout; ++o) for (int c = 0; c < C; ++c) for (int t = 0; t < T; ++t)
                mov((long)o * l.cin + (long)c * T + t, (long)o * l.cin + (long)t * C + c);
        } else {  // LIN_EXPAND: rows permuted
            const int C = l.lin_C;
            for (int c = 0; c < C; ++c) for (int t = 0; t < T; ++t) for (int k = 0; k < l.cin; ++k)
                mov(((long)c * T + t) * l.cin + k, ((long)t * C + c) * l.cin + k);
        }
    } else if (s.kind == 3) {  // v over matrix columns
        if (l.op == OP_CONV) { for (int ci = 0; ci < l.cin; ++ci) for (int j = 0; j < l.k; ++j) mov((long)ci * l.k + j, (long)j * l.cin + ci); }
        else if (l.op == OP_CONVT) { for (int ci = 0; ci < l.cin; ++ci) for (int j = 0; j < l.k; ++j) mov((long)ci * l.k + j, (long)(l.k - 1 - j) * l.cin + ci); }
        else if (l.lin_kind == LIN_HEAD) { const int C = l.lin_C; for (int c = 0; c < C; ++c) for (int t = 0; t < T; ++t) mov((long)c * T + t, (long)t * C + c); }
        else memcpy(dst, src, cnt * 4);
    } else {  // bias (0) or u (2): per output row
        if (l.op == OP_LINEAR && l.lin_kind == LIN_EXPAND) { const int C = l.lin_C; for (int c = 0; c < C; ++c) for (int t = 0; t < T; ++t) mov((long)c * T + t, (long)t * C + c); }
        else memcpy(dst, src, cnt * 4);
    }
}
static size_t entry_param_offset(const sgv_engine* e, const StateEntry& s) {
    if (s.gn >= 0) return s.kind == 4 ? e->gns[s.gn].gamma : e->gns[s.gn].beta;
    const Layer& l = e->layers[s.layer];
    switch (s.kind) { case 0: return l.b; case 1: return l.w; case 2: return l.u; default: return l.v; }
}
static size_t entry_grad_offset(const sgv_engine* e, const StateEntry& s) {
    if (!s.has_grad) return NPOS;
    if (s.gn >= 0) return s.kind == 4 ? e->gns[s.gn].ggamma : e->gns[s.gn].gbeta;
    const Layer& l = e->layers[s.layer];
    return s.kind == 0 ? l.gb : (s.kind == 1 ? l.gw : NPOS);
}
static const StateEntry* find_entry(sgv_engine* e, const char* name) {
    auto it = e->entry_index.find(name);
    if (it == e->entry_index.end()) return nullptr;
    return &e->entries[it->second];
}

static int refresh_copies(sgv_engine* e) {
    int r = opt_make_copies(e->adam_dev, e->items_copy, e->n_items_copy, e->dt, e->stream);
    if (r) return fail(SGV_ERR_HIP, "make_copies launch failed");
    e->copies_fresh = true;
    return 0;
}

static int run_sn(sgv_engine* e, int train) {
    // tmp_t of the fused layers may already hold W^T u from the last AdamW pass (still valid: neither W nor u
    // changed since); eval forwards never read or clobber it
    const bool reuse = train && e->wtu_fresh;
    const WorkItem* it1 = reuse ? e->items_sn_unf : e->items_sn;
    const int n1 = reuse ? e->n_items_sn_unf : e->n_items_sn;
    if (opt_sn_power_iteration(e->sn_dev, it1, n1, e->items_sn, e->n_items_sn, e->items_ts, e->n_items_ts, e->items_ss, e->n_items_ss,
                               (int)e->layers.size(), train, e->stream))
        return fail(SGV_ERR_HIP, "spectral-norm launch failed");
    if (train) e->wtu_fresh = false;     // u moved
    return 0;
}

static int export_act(sgv_engine* e, const Tensor& t, int B, float* host) {
    // [B][T][C] compute dtype -> [B][C][T] fp32
    const long cnt = (long)B * e->T * t.C;
    if ((size_t)cnt > e->xpose_floats) return fail(SGV_ERR_ARG, "activation too large for the export buffer");
    ew_transpose(t.f32 ? 0 : e->dt, 0, t.p, e->xpose_tmp, B, e->T, t.C, t.ld, e->T, (long)e->T * t.ld, (long)t.C * e->T, e->stream);
    HIPCHK(hipMemcpyAsync(host, e->xpose_tmp, cnt * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" {

int sgv_load_state(sgv_engine* e, const char* name, const float* host, size_t count) {
    if (!e || !name || !host) return fail(SGV_ERR_ARG, "null argument");
    const StateEntry* s = find_entry(e, name);
    if (!s) return fail(SGV_ERR_NAME, "unknown state key '%s'", name);
    if ((long)count != s->count()) return fail(SGV_ERR_ARG, "size mismatch for '%s': got %zu expected %ld", name, count, s->count());
    std::vector<float> tmp(count);
    permute_entry(e, *s, host, tmp.data(), +1);
    HIPCHK(hipMemcpy(e->params + entry_param_offset(e, *s), tmp.data(), count * 4, hipMemcpyHostToDevice));
    e->copies_fresh = false;
    e->wtu_fresh = false;
    return SGV_OK;
}

int sgv_export_state(sgv_engine* e, const char* name, float* host, size_t count) {
    if (!e || !name || !host) return fail(SGV_ERR_ARG, "null argument");
    const StateEntry* s = find_entry(e, name);
    if (!s) return fail(SGV_ERR_NAME, "unknown state key '%s'", name);
    if ((long)count != s->count()) return fail(SGV_ERR_ARG, "size mismatch for '%s'", name);
    std::vector<float> tmp(count);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(tmp.data(), e->params + entry_param_offset(e, *s), count * 4, hipMemcpyDeviceToHost));
    permute_entry(e, *s, tmp.data(), host, -1);
    return SGV_OK;
}

int sgv_export_grad(sgv_engine* e, const char* name, float* host, size_t count, int* is_none) {
    if (!e || !name || !host) return fail(SGV_ERR_ARG, "null argument");
    const StateEntry* s = find_entry(e, name);
    if (!s) return fail(SGV_ERR_NAME, "unknown state key '%s'", name);
    if ((long)count != s->count()) return fail(SGV_ERR_ARG, "size mismatch for '%s'", name);
    const size_t go = entry_grad_offset(e, *s);
    if (is_none) *is_none = (go == NPOS);
    if (go == NPOS) { memset(host, 0, count * 4); return SGV_OK; }
    std::vector<float> g(count);
    CHK(lp_sync(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(g.data(), e->grads + go, count * 4, hipMemcpyDeviceToHost));
    if (s->kind == 1) {
        // spectral-norm chain rule on the host (fp64): g_orig = (G - <G,W>/sigma * u v^T) / sigma
        const Layer& l = e->layers[s->layer];
        std::vector<float> w(count), u(l.cout), v((size_t)l.cin * l.k);
        float sig[2];
        HIPCHK(hipMemcpy(w.data(), e->params + l.w, count * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(u.data(), e->params + l.u, u.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(v.data(), e->params + l.v, v.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(sig, e->sn_sigma + 2 * l.sn, 8, hipMemcpyDeviceToHost));
        double dot = 0.0;
        for (size_t i = 0; i < count; ++i) dot += (double)g[i] * (double)w[i];
        const double c = dot / sig[0];
        for (int j = 0; j < l.k; ++j) for (int r = 0; r < l.cout; ++r) for (int cc = 0; cc < l.cin; ++cc) {
            const size_t i = ((size_t)j * l.cout + r) * l.cin + cc;
            g[i] = (float)(((double)g[i] - c * (double)u[r] * (double)v[(size_t)j * l.cin + cc]) / sig[0]);
        }
    }
    permute_entry(e, *s, g.data(), host, -1);
    return SGV_OK;
}

int sgv_export_adam(sgv_engine* e, const char* name, float* host_m, float* host_v, size_t count) {
    if (!e || !name) return fail(SGV_ERR_ARG, "null argument");
    const StateEntry* s = find_entry(e, name);
    if (!s) return fail(SGV_ERR_NAME, "unknown state key '%s'", name);
    const size_t go = entry_grad_offset(e, *s);
    if (go == NPOS) return fail(SGV_ERR_ARG, "'%s' has no optimizer state", name);
    if ((long)count != s->count()) return fail(SGV_ERR_ARG, "size mismatch for '%s'", name);
    std::vector<float> t(count);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (host_m) { HIPCHK(hipMemcpy(t.data(), e->adam_m + go, count * 4, hipMemcpyDeviceToHost)); permute_entry(e, *s, t.data(), host_m, -1); }
    if (host_v) { HIPCHK(hipMemcpy(t.data(), e->adam_v + go, count * 4, hipMemcpyDeviceToHost)); permute_entry(e, *s, t.data(), host_v, -1); }
    return SGV_OK;
}

int sgv_prepare(sgv_engine* e) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    return refresh_copies(e);
}

int sgv_set_input(sgv_engine* e, const float* x_dev, int batch) {
    if (!e || !x_dev) return fail(SGV_ERR_ARG, "null argument");
    if (batch < 1 || batch > e->maxB) return fail(SGV_ERR_ARG, "batch %d outside [1,%d]", batch, e->maxB);
    CHK(aug_join(e));
    // [B][N][T] fp32 -> [B][T][N] compute dtype
    ew_transpose(0, e->dt, x_dev, e->x_in.p, batch, e->N, e->T, e->T, e->x_in.ld, (long)e->N * e->T, (long)e->T * e->x_in.ld, e->stream);
    x_release(e);
    e->batch = batch;
    e->have_fwd = false;
    return SGV_OK;
}

int sgv_set_eps(sgv_engine* e, int site, const float* eps_dev, int batch) {
    if (!e || !eps_dev) return fail(SGV_ERR_ARG, "null argument");
    if (site < 0 || site >= e->n_st) return fail(SGV_ERR_ARG, "eps site %d outside [0,%d)", site, e->n_st);
    if (batch < 1 || batch > e->maxB) return fail(SGV_ERR_ARG, "bad batch");
    if (site == 0) {
        HIPCHK(hipMemcpyAsync(e->eps[0], eps_dev, (size_t)batch * e->Z * 4, hipMemcpyDeviceToDevice, e->stream));
    } else {
        const int C = e->dec[site];
        ew_transpose(0, 0, eps_dev, e->eps[site], batch, C, e->T, e->T, C, (long)C * e->T, (long)e->T * C, e->stream);
    }
    e->eps_set[site] = 1;
    return SGV_OK;
}

int sgv_set_shard(sgv_engine* e, int rank, int world) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (world < 1 || rank < 0 || rank >= world) return fail(SGV_ERR_ARG, "bad shard %d of %d", rank, world);
    e->shard_rank = rank; e->shard_world = world;
    return SGV_OK;
}
int sgv_seed(sgv_engine* e, uint64_t seed) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    e->seed = seed; e->draw = 0;
    return SGV_OK;
}

int sgv_set_option(sgv_engine* e, const char* key, int value) {
    if (!e || !key) return fail(SGV_ERR_ARG, "null argument");
    if (!strcmp(key, "write_xhat")) e->write_xhat = value != 0;
    else if (!strcmp(key, "use_tr")) e->use_tr = value != 0;
    else if (!strcmp(key, "dw_side_stream")) e->use_side = value != 0 && e->side != nullptr;
    else if (!strcmp(key, "ddp_early_adamw")) e->ddp_early = value != 0;
    else if (!strcmp(key, "wire_stream")) e->use_wire = value != 0 && ensure_wire(e) != nullptr;
    else if (!strcmp(key, "vendor_gemm")) { if (value) return fail(SGV_ERR_ARG, "vendor_gemm: the library GEMM back end was removed from libsgvae.so (comparator: tests/micro/vendor)"); }
    else if (!strcmp(key, "deterministic")) e->deterministic = value != 0;
    else if (!strcmp(key, "lanes")) e->use_lanes = value != 0 && e->lane2 != nullptr;          // second compute lane (schedule only: results are bitwise the same)
    else if (!strcmp(key, "grad_bf16")) {                                                           // see the member
        if (value && (e->dt != SGV_DTYPE_BF16)) return fail(SGV_ERR_ARG, "grad_bf16 needs a bf16 engine");
        CHK(lp_sync(e));
        if (value) {
            CHK(ensure_lp_mirror(e));
            // point the optimizer's table at the mirror for the classified layers (the kernel follows the pointer only when the launch
            // says so: adamw_tiles)
            bool changed = false;
            for (auto& a : e->adam_host) {
                if (a.sn < 0 || a.sn >= (int)e->layers.size()) continue;
                const Layer& l = e->layers[a.sn];
                if (!l.lp || a.g != e->grads + l.gw || a.glp) continue;
                a.glp = reinterpret_cast<const unsigned short*>(e->grads_lp) + l.gw; changed = true;
            }
            if (changed) { HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipMemcpy(e->adam_dev, e->adam_host.data(), sizeof(AdamDesc) * e->adam_host.size(), hipMemcpyHostToDevice)); }
        }
        e->grad_bf16 = value != 0;
    }
    else if (!strcmp(key, "recompute_activations")) e->recompute_act = value != 0;               // measurement only, see block_bwd
    else if (!strcmp(key, "fused_stages")) e->use_convgn = value != 0;                          // csrc/convgn.hip kernels for the small Conv -> GroupNorm -> GELU stages
    else return fail(SGV_ERR_ARG, "unknown option '%s'", key);
    return SGV_OK;
}

static int encoder_fwd(sgv_engine* e, int B, bool join_lane) {
    const int n = e->n;
    CHK(aug_join(e));
    Tensor x = e->x_in;
    for (int i = 0; i < n; ++i) {
        CHK(block_fwd(e, e->encA[i], x, B));
        if (i == 0) CHK(aug_fire(e));          // a staged next batch: built beside the short kernels from here on
        CHK(block_fwd(e, e->encR[i], e->encA[i].st.back().a, B));
        x = e->enc_h[i];
        if (i < n - 1) {
            // the xs head of this level feeds only the decoder's posterior branch, which runs on the second lane: it goes there
            // too, beside the next encoder block (in-order on that lane, so the posterior branch needs no extra wait)
            Lane2 lane(e);
            const Layer& l = e->layers[e->xs_lin[i]];
            ew_linear_head_fwd(e->dt, x.p, e->params + l.w, e->params + l.b, e->sn_sigma + 2 * l.sn + 1, e->xs_raw[i], B, l.cin, l.cout, e->colpart, e->stream);
        }
    }
    const Layer& l = e->layers[e->last_lin];
    ew_linear_head_fwd(e->dt, x.p, e->params + l.w, e->params + l.b, e->sn_sigma + 2 * l.sn + 1, e->last, B, l.cin, l.cout, e->colpart, e->stream);
    if (join_lane) lane2_join(e);
    return 0;
}

// Decoder.forward (decoder.py:170-216) from e->zlat / e->xs_raw, then the recon head + loss pass.
static int decoder_fwd(sgv_engine* e, int B, int train, int mode_fix) {
    const int n = e->n, n_st = e->n_st;
    const long M = (long)B * e->T;
    for (int s = 1; s < n_st; ++s) {
        if (!e->eps_set[s]) ew_randn(e->eps[s], M * e->dec[s], e->seed, (e->draw++) * 8 + s, e->stream, (long)e->T * e->dec[s], e->shard_world, e->shard_rank);
    }
    {
        const Layer& l = e->layers[e->start_lin];
        ew_linear_expand_fwd(e->dt, e->zlat, e->params + l.w, e->params + l.b, e->sn_sigma + 2 * l.sn + 1, e->sbuf.p, B, l.cin, l.cout, e->stream);
    }
    CHK(block_fwd(e, e->decS, e->sbuf, B));
    static const int hoist = getenv("SGV_LANE_HOIST") ? atoi(getenv("SGV_LANE_HOIST")) : 1;
    for (int i = 0; i < n_st; ++i) {
        const bool post = i < n_st - 1;
        auto xs_lift = [&]() -> int {       // xs lift of the posterior branch: depends on the encoder only
            const Layer& l = e->layers[e->xs_exp[i]];
            const int lvl = n - 2 - i;
            ew_linear_expand_fwd(e->dt, e->xs_raw[lvl], e->params + l.w, e->params + l.b, e->sn_sigma + 2 * l.sn + 1, e->xl[i].p, B, l.cin, l.cout, e->stream);
            return block_fwd(e, e->decX[i], e->xl[i], B);
        };
        if (post && hoist) {
            // hoisted onto the second lane beside this stage's up-sampling and residual blocks: the lane's chain (lift, condition_xz) was
            // twice as long as the prior branch it ran beside, and the main stream waited for it at the join
            Lane2 lane(e);
            CHK(xs_lift());
        }
        CHK(block_fwd(e, e->decU[i], e->zs[i], B));
        CHK(block_fwd(e, e->decD[i], e->decU[i].st.back().a, B));
        if (!post) break;
        const int C = e->dec[i + 1];
        {   // posterior branch (xs lift -> condition_xz) on the second lane, beside the prior branch below
            Lane2 lane(e);
            if (!hoist) CHK(xs_lift());
            CHK(block_fwd(e, e->decQ1[i], e->cat[i], B));
            CHK(block_fwd(e, e->decQ2[i], e->decQ1[i].st.back().a, B));
        }
        CHK(block_fwd(e, e->decP1[i], e->dec_out[i], B));
        CHK(block_fwd(e, e->decP2[i], e->decP1[i].st.back().a, B));
        lane2_join(e);
        ew_stage_fwd(e->dt, (const float*)e->decP2[i].st[0].y.p, (const float*)e->decQ2[i].st[0].y.p, e->eps[i + 1], e->dec_out[i].p, e->dec_out[i].ld,
                     e->zs[i + 1].p, e->zs[i + 1].ld, e->zmap[i], (int)M, C, mode_fix ? 1e-10f : 1.0f, e->scal + 3 + i, 1.0f / B, (double*)e->colpart, e->stream);
    }
    // recon head: conv -> GroupNorm stats -> tanh + loss (+ backward reductions in training)
    Stage& S = e->recon.st[0];
    const Layer& L = e->layers[S.layer];
    const GNLayer& g = e->gns[S.gn];
    GNParams p = gn_base(e, g, B);
    p.y = S.y.p; p.ldy = S.y.ld; p.sums = e->stats + S.sums; p.part = e->colpart;
    if (conv_fwd_fuses_stats(e, L, e->dec_out[n_st - 1], S.y, M, p.Cg, p.G)) {        // statistics from the GEMM epilogue: one 608 MB pass less
        CHK(conv_fwd(e, L, e->dec_out[n_st - 1], S.y, M, p.sums, p.Cg, p.G));
    } else {
        CHK(conv_fwd(e, L, e->dec_out[n_st - 1], S.y, M));
        ew_gn_stats(e->dt, p, e->stream);
    }
    p.dout = e->x_in.p; p.lddout = e->x_in.ld; p.loss_type = e->cfg.loss_type;
    p.loss_sums = e->scal;
    if (e->write_xhat || !train) { p.out = e->xhat.p; p.ldout = e->xhat.ld; }
    if (train) {
        p.sums2 = e->stats + S.sums2; p.dgamma = e->recon_unit; p.dbeta = e->recon_unit + e->N;
        p.dbias = e->recon_unit + 2L * e->N; p.gscale = 1.0f;
    }
    ew_recon_loss(e->dt, train, p, e->stream);
    return 0;
}

static int read_scalars(sgv_engine* e, int B, float* scalars_host) {
    double h[16];
    HIPCHK(hipMemcpyAsync(h, e->scal, sizeof(h), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const double numel = (double)B * e->T * e->N;
    for (int i = 0; i < SGV_MAX_SCALARS; ++i) scalars_host[i] = 0.f;
    scalars_host[0] = (float)(h[0] / numel);
    scalars_host[1] = (float)h[2];
    for (int i = 0; i + 1 < e->n_st; ++i) scalars_host[2 + i] = (float)h[3 + i];
    scalars_host[1 + e->n_st] = (float)(h[1] / numel);
    return 0;
}

int sgv_forward(sgv_engine* e, int train, int mode_fix, float* scalars_host) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (e->batch < 1) return fail(SGV_ERR_STATE, "no input set");
    if (train && mode_fix) return fail(SGV_ERR_ARG, "mode_fix is an inference path");
    if (!e->copies_fresh) CHK(refresh_copies(e));
    const int B = e->batch;
    if (!e->deterministic) HIPCHK(hipMemsetAsync(e->stats, 0, e->n_stats_fwd * 8, e->stream));   // only the fp64-atomic statistics epilogue accumulates
    HIPCHK(hipMemsetAsync(e->scal, 0, 16 * 8, e->stream));
    CHK(run_sn(e, train));
    CHK(encoder_fwd(e, B, false));
    if (!e->eps_set[0]) ew_randn(e->eps[0], (long)B * e->Z, e->seed, (e->draw++) * 8, e->stream, e->Z, e->shard_world, e->shard_rank);
    ew_latent_fwd(e->last, e->eps[0], e->zlat, B, e->Z, e->scal + 2, e->stream);
    CHK(decoder_fwd(e, B, train, mode_fix));
    e->have_fwd = true;
    e->fwd_train = train != 0;
    for (int s = 0; s < e->n_st; ++s) e->eps_set[s] = 0;
    x_release(e);                        // re-recorded at the end of the backward pass, which reads the batch again
    if (scalars_host) CHK(read_scalars(e, B, scalars_host));
    return SGV_OK;
}

int sgv_decode(sgv_engine* e, const float* z_dev, const float* xs_dev, int batch, int mode_fix, float* scalars_host) {
    if (!e || !z_dev) return fail(SGV_ERR_ARG, "null argument");
    if (batch < 1 || batch > e->maxB) return fail(SGV_ERR_ARG, "batch %d outside [1,%d]", batch, e->maxB);
    if (!e->copies_fresh) CHK(refresh_copies(e));
    const int B = batch;
    e->batch = B;
    if (!e->deterministic) HIPCHK(hipMemsetAsync(e->stats, 0, e->n_stats_fwd * 8, e->stream));   // only the fp64-atomic statistics epilogue accumulates
    HIPCHK(hipMemsetAsync(e->scal, 0, 16 * 8, e->stream));
    CHK(run_sn(e, 0));
    HIPCHK(hipMemcpyAsync(e->zlat, z_dev, (size_t)B * e->Z * 4, hipMemcpyDeviceToDevice, e->stream));
    if (xs_dev) {
        // list order of Encoder.forward's return: [xs_{n-2}, ..., xs_0]
        for (int j = 0; j < e->n - 1; ++j)
            HIPCHK(hipMemcpyAsync(e->xs_raw[e->n - 2 - j], xs_dev + (size_t)j * B * e->H, (size_t)B * e->H * 4, hipMemcpyDeviceToDevice, e->stream));
    } else {
        return fail(SGV_ERR_ARG, "sgv_decode needs xs (Decoder.forward with xs=None leaves z unchanged between stages in the reference; not supported)");
    }
    CHK(decoder_fwd(e, B, 0, mode_fix));
    e->have_fwd = true;
    e->fwd_train = false;
    for (int s = 0; s < e->n_st; ++s) e->eps_set[s] = 0;
    if (scalars_host) CHK(read_scalars(e, B, scalars_host));
    return SGV_OK;
}

int sgv_encode(sgv_engine* e, float* mu_host, float* logvar_host, float* xs_host) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (e->batch < 1) return fail(SGV_ERR_STATE, "no input set");
    if (!e->copies_fresh) CHK(refresh_copies(e));
    const int B = e->batch;
    if (!e->deterministic) HIPCHK(hipMemsetAsync(e->stats, 0, e->n_stats_fwd * 8, e->stream));   // only the fp64-atomic statistics epilogue accumulates
    CHK(run_sn(e, 0));
    CHK(encoder_fwd(e, B, true));
    x_release(e);
    std::vector<float> last((size_t)B * 2 * e->Z);
    HIPCHK(hipMemcpyAsync(last.data(), e->last, last.size() * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int b = 0; b < B; ++b) {
        if (mu_host) memcpy(mu_host + (size_t)b * e->Z, &last[(size_t)b * 2 * e->Z], e->Z * 4);
        if (logvar_host) memcpy(logvar_host + (size_t)b * e->Z, &last[(size_t)b * 2 * e->Z + e->Z], e->Z * 4);
    }
    if (xs_host) {
        // list order of Encoder.forward's return: [xs_{n-2}, ..., xs_0]
        for (int j = 0; j < e->n - 1; ++j)
            HIPCHK(hipMemcpy(xs_host + (size_t)j * B * e->H, e->xs_raw[e->n - 2 - j], (size_t)B * e->H * 4, hipMemcpyDeviceToHost));
    }
    return SGV_OK;
}

int sgv_get_xhat(sgv_engine* e, float* xhat_dev) {
    if (!e || !xhat_dev) return fail(SGV_ERR_ARG, "null argument");
    if (!e->have_fwd) return fail(SGV_ERR_STATE, "no forward pass to read from");
    ew_transpose(e->dt, 0, e->xhat.p, xhat_dev, e->batch, e->T, e->N, e->xhat.ld, e->T, (long)e->T * e->xhat.ld, (long)e->N * e->T, e->stream);
    return SGV_OK;
}

int sgv_get_activation(sgv_engine* e, const char* name, float* host, size_t count) {
    if (!e || !name || !host) return fail(SGV_ERR_ARG, "null argument");
    const int B = e->batch;
    std::string s(name);
    if (s == "x_in") {     // the input batch as the engine holds it (after sgv_set_input / sgv_augment_collate)
        CHK(aug_join(e));
        if (B < 1) return fail(SGV_ERR_STATE, "no input batch");
        if ((long)count != (long)B * e->T * e->N) return fail(SGV_ERR_ARG, "size mismatch for activation 'x_in'");
        return export_act(e, e->x_in, B, host);
    }
    if (!e->have_fwd) return fail(SGV_ERR_STATE, "no forward pass to read from");
    if (s.rfind("eps", 0) == 0 && s.size() == 4) {      // the noise of the last forward: eps0 [B][latent], eps{i} [B*T][C_i] (channels-last rows)
        const int site = s[3] - '0';
        if (site < 0 || site >= e->n_st) return fail(SGV_ERR_NAME, "bad noise site");
        const long want = site == 0 ? (long)B * e->Z : (long)B * e->T * e->dec[site];
        if ((long)count != want) return fail(SGV_ERR_ARG, "size mismatch for activation '%s': got %zu expected %ld", name, count, want);
        HIPCHK(hipMemcpy(host, e->eps[site], count * 4, hipMemcpyDeviceToHost));
        return SGV_OK;
    }
    auto chk = [&](long want) { return (long)count == want ? 0 : fail(SGV_ERR_ARG, "size mismatch for activation '%s': got %zu expected %ld", name, count, want); };
    auto idx = [&](const char* pre) { return atoi(s.c_str() + strlen(pre)); };
    HIPCHK(hipStreamSynchronize(e->stream));
    if (s.rfind("enc_h", 0) == 0) { int i = idx("enc_h"); if (i < 0 || i >= e->n) return fail(SGV_ERR_NAME, "bad index"); CHK(chk((long)B * e->T * e->enc[i])); return export_act(e, e->enc_h[i], B, host); }
    if (s.rfind("dec_out", 0) == 0) { int i = idx("dec_out"); if (i < 0 || i >= e->n_st) return fail(SGV_ERR_NAME, "bad index"); CHK(chk((long)B * e->T * e->dec[i + 1])); return export_act(e, e->dec_out[i], B, host); }
    if (s.rfind("zmap", 0) == 0) {
        int i = idx("zmap"); if (i < 0 || i + 1 >= e->n_st) return fail(SGV_ERR_NAME, "bad index");
        Tensor t; t.p = e->zmap[i]; t.C = e->dec[i + 1]; t.ld = t.C; t.f32 = true;
        CHK(chk((long)B * e->T * t.C)); return export_act(e, t, B, host);
    }
    if (s == "x_hat") { CHK(chk((long)B * e->T * e->N)); return export_act(e, e->xhat, B, host); }
    if (s == "mu" || s == "log_var") {
        CHK(chk((long)B * e->Z));
        std::vector<float> last((size_t)B * 2 * e->Z);
        HIPCHK(hipMemcpy(last.data(), e->last, last.size() * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) memcpy(host + (size_t)b * e->Z, &last[(size_t)b * 2 * e->Z + (s == "mu" ? 0 : e->Z)], e->Z * 4);
        return SGV_OK;
    }
    if (s == "z") { CHK(chk((long)B * e->Z)); HIPCHK(hipMemcpy(host, e->zlat, count * 4, hipMemcpyDeviceToHost)); return SGV_OK; }
    if (s.rfind("xs", 0) == 0) {
        int j = idx("xs"); if (j < 0 || j >= e->n - 1) return fail(SGV_ERR_NAME, "bad index");
        CHK(chk((long)B * e->H)); HIPCHK(hipMemcpy(host, e->xs_raw[e->n - 2 - j], count * 4, hipMemcpyDeviceToHost)); return SGV_OK;
    }
    return fail(SGV_ERR_NAME, "unknown activation '%s'", name);
}

// ---- RCCL, resolved at run time --------------------------------------------------------------------------------
namespace {
struct RcclApi {
    struct Id128 { char b[128]; };             // ncclUniqueId: 128 opaque bytes, passed by value
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
const int kNcclFloat32 = 7, kNcclBfloat16 = 9, kNcclAvg = 4;      // ncclDataType_t / ncclRedOp_t values of rccl.h (NCCL >= 2.10 ABI)
int rccl_load() {
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(SGV_ERR_STATE, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce)
        return fail(SGV_ERR_STATE, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
    g_rccl.h = h;
    return 0;
}
// ---- test double for the collective (sgv_test_fake_collective): "all-reduce" = multiply the range in place by k on the given
// stream.  With k a power of two every element the engine hands to a collective is scaled exactly, so a step through the fake must
// leave bitwise the state of a plain step at (k alpha, k beta) -- if and only if every gradient element and every <G,W> slot went
// through exactly one collective (backward is linear in (alpha, beta); tests/test_modules_gpu.py).
float g_fake_k = 0.f;
long g_fake_calls = 0, g_fake_elems = 0;
__global__ void fake_scale_f32_kernel(float* p, size_t n, float k) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] *= k;
}
__global__ void fake_scale_bf16_kernel(bf16_t* p, size_t n, float k) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (bf16_t)((float)p[i] * k);
}
int fake_allreduce(const void* in, void* out, size_t count, int dtype, int op, void* comm, hipStream_t st) {
    (void)comm;
    if (in != out || op != kNcclAvg || (dtype != kNcclFloat32 && dtype != kNcclBfloat16)) return 1;
    ++g_fake_calls; g_fake_elems += (long)count;
    if (!count) return 0;
    const int blocks = (int)std::min<size_t>((count + 255) / 256, 4096);
    if (dtype == kNcclFloat32) hipLaunchKernelGGL(fake_scale_f32_kernel, dim3(blocks), dim3(256), 0, st, (float*)out, count, g_fake_k);
    else hipLaunchKernelGGL(fake_scale_bf16_kernel, dim3(blocks), dim3(256), 0, st, (bf16_t*)out, count, g_fake_k);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int rccl_fail(const char* what, int rc) {
    return fail(SGV_ERR_HIP, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}
}  // namespace

int sgv_rccl_probe(void) { return rccl_load(); }
int sgv_rccl_comm_count(void* comm, int* nranks) {
    if (!comm || !nranks) return fail(SGV_ERR_ARG, "null argument");
    CHK(rccl_load());
    if (g_fake_k != 0.f) { *nranks = 0; return SGV_OK; }                  // the test double has no ranks
    if (!g_rccl.CommCount) return fail(SGV_ERR_STATE, "librccl lacks ncclCommCount");
    const int rc = g_rccl.CommCount(comm, nranks);
    return rc ? rccl_fail("ncclCommCount", rc) : SGV_OK;
}
int sgv_test_fake_collective(float k, long* calls, long* elems) {
    if (calls) *calls = g_fake_calls;
    if (elems) *elems = g_fake_elems;
    g_fake_calls = 0; g_fake_elems = 0;
    if (k != 0.f) {
        g_fake_k = k;
        g_rccl.AllReduce = fake_allreduce;
        if (!g_rccl.h) g_rccl.h = (void*)&g_fake_k;                     // rccl_load: nothing to resolve while the double is installed
    } else if (g_fake_k != 0.f) {
        const bool own = g_rccl.h == (void*)&g_fake_k;
        g_fake_k = 0.f;
        if (own) g_rccl = RcclApi();                                    // the next rccl_load resolves the real library
        else g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(g_rccl.h, "ncclAllReduce");
    }
    return SGV_OK;
}
int sgv_rccl_unique_id(void* id128) {
    if (!id128) return fail(SGV_ERR_ARG, "null argument");
    CHK(rccl_load());
    const int rc = g_rccl.GetUniqueId(id128);
    return rc ? rccl_fail("ncclGetUniqueId", rc) : SGV_OK;
}
static std::map<void*, int> g_comm_ranks;      // communicator -> number of ranks (one rank: the mean is the identity, nothing is issued)
// a one-rank communicator exchanges nothing: its all-reduces are skipped -- unless SGV_FORCE_COLLECTIVE=1 asks for the one-GPU
// rehearsal of the N > 1 path (every bucket packed, handed to ncclAllReduce and unpacked as with more ranks)
static bool comm_is_single(void* comm) {
    const char* f = getenv("SGV_FORCE_COLLECTIVE");      // read per call: tests switch it inside one process
    if (f && atoi(f) == 1) return false;
    auto it = g_comm_ranks.find(comm);
    return it != g_comm_ranks.end() && it->second == 1;
}
int sgv_rccl_comm_init(void** comm_out, int nranks, const void* id128, int rank) {
    if (!comm_out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(SGV_ERR_ARG, "bad argument");
    CHK(rccl_load());
    RcclApi::Id128 id;
    memcpy(id.b, id128, 128);
    const int rc = g_rccl.CommInitRank(comm_out, nranks, id, rank);
    if (!rc) g_comm_ranks[*comm_out] = nranks;
    return rc ? rccl_fail("ncclCommInitRank", rc) : SGV_OK;
}
int sgv_rccl_allreduce(void* comm, void* dev_buf, size_t count, int dtype, void* stream) {
    if (!comm || !dev_buf) return fail(SGV_ERR_ARG, "null argument");
    if (dtype != SGV_DTYPE_F32 && dtype != SGV_DTYPE_BF16) return fail(SGV_ERR_ARG, "bad dtype");
    CHK(rccl_load());
    const int rc = g_rccl.AllReduce(dev_buf, dev_buf, count, dtype == SGV_DTYPE_BF16 ? kNcclBfloat16 : kNcclFloat32, kNcclAvg, comm, (hipStream_t)stream);
    return rc ? rccl_fail("ncclAllReduce", rc) : SGV_OK;
}
int sgv_rccl_comm_destroy(void* comm) {
    if (!comm) return SGV_OK;
    CHK(rccl_load());
    const int rc = g_rccl.CommDestroy(comm);
    return rc ? rccl_fail("ncclCommDestroy", rc) : SGV_OK;
}
// bucket b: wait (on the communication stream) for what the engine stream holds so far, average it over the ranks.
// split_dots: the <G,W_eff> slots of a weight bucket's conv layers travel with the bucket (a second, tiny fp32 all-reduce of
// bucket_dots[b]) and the small bucket leaves them out -- the bucket's AdamW then needs nothing from the end of backward.
static int rccl_bucket(sgv_engine* e, void* comm, hipStream_t cs, int b, hipEvent_t done, bool split_dots = false) {
    hipEvent_t ev = next_event(e);
    if (!ev) return fail(SGV_ERR_HIP, "event creation failed");
    HIPCHK(hipEventRecord(ev, e->stream));
    HIPCHK(hipStreamWaitEvent(cs, ev, 0));
    float* g = e->grads + e->buckets[b].first;
    if (!comm_is_single(comm)) {
        const bool small = b == (int)e->buckets.size() - 1;
        const bool lp = b < (int)e->bucket_packed.size() && e->bucket_packed[b];
        void* w = lp ? (void*)((char*)e->grads_lp + 2 * e->buckets[b].first) : (void*)g;
        size_t cnt = e->buckets[b].second;
        if (small && split_dots) { w = (void*)(g + e->dots_total); cnt -= e->dots_total; }
        int rc = g_rccl.AllReduce(w, w, cnt, lp ? kNcclBfloat16 : kNcclFloat32, kNcclAvg, comm, cs);
        if (rc) return rccl_fail("ncclAllReduce", rc);
        if (!small && split_dots && e->bucket_dots[b].second) {
            float* d = e->grads + e->bucket_dots[b].first;
            rc = g_rccl.AllReduce(d, d, e->bucket_dots[b].second, kNcclFloat32, kNcclAvg, comm, cs);
            if (rc) return rccl_fail("ncclAllReduce(<G,W> slots)", rc);
        }
    }
    if (done) HIPCHK(hipEventRecord(done, cs));
    return 0;
}
int sgv_allreduce_grads(sgv_engine* e, void* rccl_comm, void* comm_stream) {
    if (!e || !rccl_comm) return fail(SGV_ERR_ARG, "null argument");
    CHK(rccl_load());
    CHK(join_side(e));
    const hipStream_t cs = comm_stream ? (hipStream_t)comm_stream : e->stream;
    for (int b = 0; b < (int)e->buckets.size(); ++b) CHK(rccl_bucket(e, rccl_comm, cs, b, nullptr));
    if (cs != e->stream) {
        hipEvent_t ev = next_event(e);
        if (!ev) return fail(SGV_ERR_HIP, "event creation failed");
        HIPCHK(hipEventRecord(ev, cs));
        HIPCHK(hipStreamWaitEvent(e->stream, ev, 0));
    }
    return SGV_OK;
}
int sgv_set_rccl(sgv_engine* e, void* rccl_comm, void* comm_stream) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (rccl_comm && e->cb) return fail(SGV_ERR_STATE, "a bucket callback is registered: use one of the two data-parallel paths");
    if (rccl_comm) {
        if (!comm_stream) return fail(SGV_ERR_ARG, "sgv_set_rccl needs a communication stream of its own");
        CHK(rccl_load());
        while (e->bucket_done.size() < e->buckets.size()) {
            hipEvent_t ev = nullptr;
            HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->bucket_done.push_back(ev);
        }
        e->bucket_pending.assign(e->buckets.size(), 0);
        if (!e->tn_sched) HIPCHK(hipMalloc((void**)&e->tn_sched, 8 * 520 * sizeof(int)));
    }
    e->comm = rccl_comm;
    e->comm_stream = (hipStream_t)comm_stream;
    return SGV_OK;
}

int sgv_set_bucket_callback(sgv_engine* e, sgv_bucket_cb cb, void* user) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (cb && e->comm) return fail(SGV_ERR_STATE, "an RCCL communicator is registered: use one of the two data-parallel paths");
    if (cb && !e->tn_sched) HIPCHK(hipMalloc((void**)&e->tn_sched, 8 * 520 * sizeof(int)));
    e->cb = cb; e->cb_user = user;
    return SGV_OK;
}
int sgv_grad_buffer(sgv_engine* e, float** dev_ptr, size_t* count_elems) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (dev_ptr) *dev_ptr = e->grads;
    if (count_elems) *count_elems = e->n_grads;
    return SGV_OK;
}
int sgv_scale_grads(sgv_engine* e, float factor) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    ew_scale(e->grads, factor, (long)e->n_grads, e->stream);
    return SGV_OK;
}

static int adamw_begin(sgv_engine* e);
static int adamw_finish(sgv_engine* e);
static int adamw_range(sgv_engine* e, float lr, int bucket_lo, int bucket_hi, int which, hipStream_t st);
static int adamw_bucket_async(sgv_engine* e, float lr, int b, hipStream_t st);
static int adamw_tiles(sgv_engine* e, float lr, int t0, int t1, hipStream_t st, bool from_lp);
int sgv_adamw_step(sgv_engine* e, float lr);
// fuse_lr >= 0: also run the optimizer, and start the AdamW of every conv-weight bucket on the side stream as soon as
// that bucket's gradients are final, under the rest of backward (single-GPU path: no bucket callback registered)
static int backward_impl(sgv_engine* e, float alpha, float beta, float fuse_lr) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (!e->have_fwd || !e->fwd_train) return fail(SGV_ERR_STATE, "sgv_backward needs a preceding sgv_forward(train=1)");
    const bool fuse = fuse_lr >= 0.f;
    static const int early_on = getenv("SGV_EARLY_ADAM") ? atoi(getenv("SGV_EARLY_ADAM")) : 1;
    const bool early = early_on && fuse && !e->cb && !e->comm && e->side && e->use_side && !e->timing;
    // engine-issued collectives with the learning rate in hand (sgv_backward_step on a registered communicator): every weight
    // bucket's <G,W> slots are averaged with the bucket and its conv-weight AdamW starts on the optimizer stream as soon as both
    // have landed, under the rest of backward -- the data-parallel mirror of `early`
    const bool dearly = e->ddp_early && fuse && e->comm && !comm_is_single(e->comm) && !e->timing && ensure_opt(e);
    if (early || dearly) CHK(adamw_begin(e));
    // chunked exchange of the last weight bucket: it must be exactly the first encoder layer's tiled weight
    const int last_b = (int)e->buckets.size() - 2;
    const int L0i = e->encA[0].st[0].layer;
    bool last_chunked = false;
    std::vector<hipEvent_t> chunk_done;
    if (dearly && e->ddp_last_chunks > 1 && last_b >= 0) {
        const Layer& L0 = e->layers[L0i];
        const int rt6 = (L0.cout + 63) / 64, ct6 = (L0.cin + 63) / 64;
        last_chunked = e->encA[0].st.size() == 1 && L0.k == 1 && L0.has_grad && layer_fused_adam(L0) && L0.cout % (128 * e->ddp_last_chunks) == 0 &&
                       e->buckets[last_b].first == L0.gw && e->buckets[last_b].second == align_up((size_t)L0.nw(), 4) &&
                       e->tile_off[last_b + 1] - e->tile_off[last_b] == rt6 * ct6 && e->flat_off[last_b + 1] == e->flat_off[last_b] &&
                       2.0e-9 * (double)((long)e->batch * e->T) * L0.cout * L0.cin > e->ddp_chunk_min_gf;      // the big-GEMM regime: main stream, split-K 1
    }
    const int B = e->batch, n = e->n, n_st = e->n_st;
    const long M = (long)B * e->T;
    const float coefB = beta / (float)B;
    int bucket = 0;
    e->ev_next = 0;
    e->recompute_bytes = 0;
    int early_err = 0;
    // the inputs of a bucket's collective are what the main stream and the side stream hold so far.  They are gathered on the
    // stream the collective is issued from (the communicator's stream; with a callback the wire stream, option "wire_stream"), and
    // the bf16 wire copy is packed THERE: the main stream neither waits for the side stream's weight-gradient GEMMs nor runs the
    // pack pass (0.45 ms per step at preset 1).  Without a wire stream (a caller that orders itself after the engine stream) the
    // main stream joins the side stream and packs, as before.
    auto gather_on = [&](hipStream_t t) -> int {
        hipEvent_t ev = next_event(e);
        if (!ev || hipEventRecord(ev, e->stream) != hipSuccess || hipStreamWaitEvent(t, ev, 0) != hipSuccess) return 1;
        if (e->side_dirty) {
            hipEvent_t ev2 = next_event(e);
            if (!ev2 || hipEventRecord(ev2, e->side) != hipSuccess || hipStreamWaitEvent(t, ev2, 0) != hipSuccess) return 1;
        }
        return 0;
    };
    auto fire_at = [&](int b) {
        if (b < 0 || b >= (int)e->buckets.size()) return;
        if (e->cb || (e->comm && !comm_is_single(e->comm))) e->coll_inflight = true;
        if (e->comm || e->cb) {
            const hipStream_t ws = e->comm ? e->comm_stream : (e->use_wire ? e->wire : nullptr);
            if (ws ? gather_on(ws) : join_side(e)) { early_err = 1; return; }
            if (e->payload_bf16 && b != (int)e->buckets.size() - 1 && !(e->comm && comm_is_single(e->comm))) {
                pack_bucket(e, b, ws ? ws : e->stream);
                e->bucket_packed[b] = 3;
            }
        }
        if (e->comm) {
            if (rccl_bucket(e, e->comm, e->comm_stream, b, e->bucket_done[b], dearly)) { early_err = 1; return; }
            e->bucket_pending[b] = 1;
            if (dearly && b < (int)e->buckets.size() - 2) {
                if (hipStreamWaitEvent(e->opt, e->bucket_done[b], 0) != hipSuccess) { early_err = 1; return; }
                e->bucket_pending[b] = 0;
                if (adamw_bucket_async(e, fuse_lr, b, e->opt)) early_err = 1;
            }
        } else if (e->cb) {
            e->cb(e->cb_user, b, e->buckets[b].first, e->buckets[b].second);
        } else if (early && b < (int)e->buckets.size() - 2) {
            // the bucket's weight gradients (and the <G,W> slots of its conv layers) are final once everything enqueued
            // so far has run: AdamW of its conv weights goes to the side stream, under the remaining backward
            hipEvent_t ev = next_event(e);
            if (!ev || hipEventRecord(ev, e->stream) != hipSuccess || hipStreamWaitEvent(e->side, ev, 0) != hipSuccess) { early_err = 1; return; }
            if (adamw_range(e, fuse_lr, b, b + 1, 1, e->side)) early_err = 1;
            e->side_dirty = true;
        }
    };
    // fixed-order sums of the partials collected so far: the <G,W_eff> scalars of the layers whose dY kernels have been enqueued
    // (every fire point: the bucket's AdamW / all-reduce reads them), the GroupNorm affine and bias gradients (small bucket)
    auto flush_fin = [&](bool affine) {
        if (!e->fin_dots.empty()) { ew_fin_dots(e->fin_dots.data(), (int)e->fin_dots.size(), e->stream); e->fin_dots.clear(); }
        if (affine && !e->fin_affine.empty()) { ew_fin_affine(e->fin_affine.data(), (int)e->fin_affine.size(), e->stream); e->fin_affine.clear(); }
    };
    e->fin_dots.clear(); e->fin_affine.clear();
    // <G,W_eff> of the Linear layers is computed from G itself: in one launch at the end of backward -- unless a collective may
    // already be reducing a released bucket's gradients in place by then (fp32 wire format): with a callback or a communicator
    // every bucket's Linear layers get theirs at the bucket's fire point.  Same per-item partials, same fixed-order sums.
    const bool dots_per_bucket = e->cb || e->comm;
    int lin_err = 0;
    auto lin_dots = [&](int b0, int b1) {
        const int d0 = e->dot_off[b0], d1 = e->dot_off[b1];
        if (d1 > d0 && opt_sn_grad_dot(e->sn_dev, e->items_dot + d0, d1 - d0, e->lin_dot_part + d0, e->stream)) lin_err = 1;
        e->fin_dots.insert(e->fin_dots.end(), e->fin_lin_dots.begin() + e->fin_lin_off[b0], e->fin_lin_dots.begin() + e->fin_lin_off[b1]);
    };
    auto fire = [&]() { if (dots_per_bucket) lin_dots(bucket, bucket + 1); flush_fin(false); fire_at(bucket); ++bucket; };
    const int small_bucket = (int)e->buckets.size() - 1;
    // no zero-fills: every gradient of the small zone (biases, GroupNorm affine, <G,W_eff> slot 0) and every backward group sum
    // is written, not accumulated, by its fixed-order reduction; tensors that get no gradient stay at their initial zero
    // ---- recon head ----
    {
        Stage& S = e->recon.st[0];
        const Layer& L = e->layers[S.layer];
        const GNLayer& g = e->gns[S.gn];
        const float gs = alpha / (float)((double)M * e->N);
        GNParams p = gn_base(e, g, B);
        p.y = S.y.p; p.ldy = S.y.ld; p.sums = e->stats + S.sums; p.sums2 = e->stats + S.sums2;
        p.dout = e->x_in.p; p.lddout = e->x_in.ld; p.loss_type = e->cfg.loss_type; p.gscale = gs;
        p.out = e->dy_recon.p; p.ldout = e->dy_recon.ld;
        p.cdot = e->grads + L.gdot; p.cbias = e->params + L.b;
        p.cdot_part = e->red + L.dot_part; p.cdot_blocks = &e->dot_counts[0];
        ew_recon_bwd_apply(e->dt, p, e->stream);
        e->fin_dots.push_back({p.cdot_part, p.cdot, e->dot_counts[0], 0});
        ew_scale3(e->grads + g.ggamma, e->grads + g.gbeta, e->grads + L.gb, e->recon_unit, gs, e->N, e->stream);
        CHK(conv_bwd_dw(e, L, e->dy_recon, e->dec_out[n_st - 1], M));
        CHK(conv_bwd_dx(e, L, e->dy_recon, e->d_out[n_st - 1], nullptr, M));
        fire();
    }
    // ---- decoder stages ----
    static const int hoist_b = getenv("SGV_LANE_HOIST") ? atoi(getenv("SGV_LANE_HOIST")) : 1;
    for (int i = n_st - 1; i >= 0; --i) {
        const int C = e->dec[i + 1];
        if (i < n_st - 1) {
            ew_stage_bwd(e->dt, (const float*)e->decP2[i].st[0].y.p, (const float*)e->decQ2[i].st[0].y.p, e->eps[i + 1], e->dzs[i + 1].p, e->dzs[i + 1].ld,
                         e->gp[i].p, e->gq[i].p, (int)M, C, coefB, e->stream);
            Tensor d_xs = e->dcat[i]; d_xs.C = C;
            Tensor d_oq = e->dcat[i]; d_oq.C = C; d_oq.p = (char*)d_oq.p + (size_t)C * e->esz;
            auto xs_lift_bwd = [&]() -> int {
                const int r = block_bwd(e, e->decX[i], e->xl[i], d_xs, &e->d_xl[i], B);
                if (r) return r;
                const Layer& l = e->layers[e->xs_exp[i]];
                const int lvl = n - 2 - i;
                ew_linear_expand_bwd(e->dt, e->d_xl[i].p, e->xs_raw[lvl], e->params + l.w, e->sn_sigma + 2 * l.sn + 1, e->d_xs_raw[lvl],
                                     e->grads + l.gw, e->grads + l.gb, B, l.cin, l.cout, e->stream);
                return 0;
            };
            {   // posterior branch on the second lane, beside the prior branch below (they meet in the add3 after the join)
                Lane2 lane(e);
                bool q_ready = false;      // condition_xz: the output convolution's input gradient went straight into the residual block's GroupNorm backward
                CHK(block_bwd(e, e->decQ2[i], e->decQ1[i].st.back().a, e->gq[i], &e->d_qres[i], B, nullptr, &e->decQ1[i].st.back(), &q_ready, false, 0.1f));
                CHK(block_bwd(e, e->decQ1[i], e->cat[i], e->d_qres[i], &e->dcat[i], B, nullptr, nullptr, nullptr, q_ready));
                if (!hoist_b) CHK(xs_lift_bwd());
            }
            bool p_ready = false;
            CHK(block_bwd(e, e->decP2[i], e->decP1[i].st.back().a, e->gp[i], &e->d_pres[i], B, nullptr, &e->decP1[i].st.back(), &p_ready, false, 0.1f));
            CHK(block_bwd(e, e->decP1[i], e->dec_out[i], e->d_pres[i], &e->d_outp[i], B, nullptr, nullptr, nullptr, p_ready));
            lane2_join(e);
            ew_add3(e->dt, e->d_outp[i].p, e->d_outp[i].ld, e->dzs[i + 1].p, e->dzs[i + 1].ld, d_oq.p, d_oq.ld, e->d_out[i].p, e->d_out[i].ld, (int)M, C, e->stream);
            if (hoist_b) {
                // the xs lift's backward needs d_xs only: deferred onto the lane beside the residual / up-sampling blocks' backward
                // below (the mirror of the forward hoist); joined before the stage's bucket is released
                Lane2 lane(e);
                CHK(xs_lift_bwd());
            }
        }
        CHK(block_bwd(e, e->decD[i], e->decU[i].st.back().a, e->d_out[i], &e->d_u[i], B));
        CHK(block_bwd(e, e->decU[i], e->zs[i], e->d_u[i], &e->dzs[i], B));
        if (i < n_st - 1 && hoist_b) lane2_join(e);
        if (i == 0) {
            CHK(block_bwd(e, e->decS, e->sbuf, e->dzs[0], &e->d_sbuf, B));
            const Layer& l = e->layers[e->start_lin];
            ew_linear_expand_bwd(e->dt, e->d_sbuf.p, e->zlat, e->params + l.w, e->sn_sigma + 2 * l.sn + 1, e->d_z, e->grads + l.gw, e->grads + l.gb,
                                 B, l.cin, l.cout, e->stream);
        }
        fire();
    }
    // ---- latent + encoder ----
    ew_latent_bwd(e->last, e->eps[0], e->d_z, e->d_last, B, e->Z, coefB, e->stream);
    {
        const Layer& l = e->layers[e->last_lin];
        ew_linear_head_bwd(e->dt, e->d_last, e->enc_h[n - 1].p, e->params + l.w, e->sn_sigma + 2 * l.sn + 1, nullptr, e->d_h[n - 1].p,
                           e->grads + l.gw, e->grads + l.gb, B, l.cin, l.cout, e->stream);
    }
    for (int i = n - 1; i >= 0; --i) {
        const Layer& xl = e->layers[e->xs_lin[i]];
        if (xl.has_grad) {
            ew_linear_head_bwd(e->dt, e->d_xs_raw[i], e->enc_h[i].p, e->params + xl.w, e->sn_sigma + 2 * xl.sn + 1, e->d_h[i].p, e->d_h[i].p,
                               e->grads + xl.gw, e->grads + xl.gb, B, xl.cin, xl.cout, e->stream);
        }
        bool a_ready = false;       // the residual block's input gradient went straight into the GroupNorm backward of encA[i]'s last stage
        CHK(block_bwd(e, e->encR[i], e->encA[i].st.back().a, e->d_h[i], &e->enc_a_dummy[i], B, nullptr, &e->encA[i].st.back(), &a_ready));
        const Tensor x_prev = i == 0 ? e->x_in : e->enc_h[i - 1];
        if (i == 0) {
            fire();   // everything but the first block's weight gradients is now enqueued
            // <G,W_eff> of the (small) Linear layers from their weights; conv layers accumulated theirs in the dY kernels
            if (!dots_per_bucket) lin_dots(0, (int)e->buckets.size());
            if (lin_err) return fail(SGV_ERR_HIP, "grad-dot launch failed");
            // the small zone (biases, GroupNorm affine, <G,W_eff> scalars) is complete once the first conv's dY exists:
            // release it BEFORE the first-layer weight-gradient GEMM so that its all-reduce (and, with it, the AdamW
            // of every other layer) does not queue behind the 390 MB first-layer bucket
            const std::function<void()> early = [&]() { flush_fin(true); fire_at(small_bucket); };
            if (last_chunked) {
                e->dw_chunks = e->ddp_last_chunks; e->dw_chunk_layer = L0i;
                e->dw_chunk_hook = [&](int c, int n_c, int co0, int co1) -> int {
                    const Layer& L0 = e->layers[L0i];
                    const size_t off = L0.gw + (size_t)co0 * L0.cin, cnt = (size_t)(co1 - co0) * L0.cin;
                    const hipStream_t cs = e->comm_stream;
                    if (gather_on(cs)) return 1;                                  // the chunk's GEMM (main stream)
                    const bool lp = e->payload_bf16 != 0;
                    void* w = lp ? (void*)((char*)e->grads_lp + 2 * off) : (void*)(e->grads + off);
                    if (lp && !e->dw_chunk_direct) ew_pack_bf16(e->grads + off, w, (long)cnt, cs);
                    if (g_rccl.AllReduce(w, w, cnt, lp ? kNcclBfloat16 : kNcclFloat32, kNcclAvg, e->comm, cs)) return 1;
                    if (c == 0 && e->bucket_dots[last_b].second) {
                        float* d = e->grads + e->bucket_dots[last_b].first;
                        if (g_rccl.AllReduce(d, d, e->bucket_dots[last_b].second, kNcclFloat32, kNcclAvg, e->comm, cs)) return 1;
                    }
                    hipEvent_t ev = c == n_c - 1 ? e->bucket_done[last_b] : next_event(e);
                    if (!ev || hipEventRecord(ev, cs) != hipSuccess) return 1;
                    chunk_done.push_back(ev);
                    return 0;
                };
            }
            const int br = block_bwd(e, e->encA[0], x_prev, e->enc_a_dummy[0], nullptr, B, &early, nullptr, nullptr, a_ready);
            e->dw_chunks = 1; e->dw_chunk_layer = -1; e->dw_chunk_hook = nullptr;
            CHK(br);
            if (last_chunked && (int)chunk_done.size() == e->ddp_last_chunks) {
                // the chunks' updates go to the MAIN stream, behind the last GEMM chunk: it has nothing else left to do, and chunk c's
                // AdamW then runs beside chunk c + 1's exchange instead of in front of it on the communication stream's queue
                const Layer& L0 = e->layers[L0i];
                const int n_c = e->ddp_last_chunks, rows = L0.cout / n_c, ct6 = (L0.cin + 63) / 64;
                const bool lp = e->payload_bf16 != 0;
                for (int c = 0; c < n_c; ++c) {
                    const size_t off = L0.gw + (size_t)c * rows * L0.cin, cnt = (size_t)rows * L0.cin;
                    HIPCHK(hipStreamWaitEvent(e->stream, chunk_done[c], 0));
                    if (lp && !e->lp_direct) ew_unpack_bf16((const char*)e->grads_lp + 2 * off, e->grads + off, (long)cnt, e->stream);
                    CHK(adamw_tiles(e, fuse_lr, e->tile_off[last_b] + (c * rows / 64) * ct6, e->tile_off[last_b] + ((c + 1) * rows / 64) * ct6, e->stream, lp && e->lp_direct));
                }
                e->bucket_updated[last_b] = 1;
            }
        } else {
            CHK(block_bwd(e, e->encA[i], x_prev, e->enc_a_dummy[i], &e->d_h[i - 1], B, nullptr, nullptr, nullptr, a_ready));
        }
    }
    if (last_chunked && e->bucket_updated[last_b]) ++bucket;      // exchanged and updated chunk by chunk above
    else fire();                                                   // first encoder block's weights
    if (early_err) return fail(SGV_ERR_HIP, "early AdamW launch failed");
    if (fuse) {
        const int nbk = (int)e->buckets.size();
        if (early) {
            CHK(join_side(e));                                            // side-stream dW GEMMs of the last bucket
            CHK(adamw_range(e, fuse_lr, nbk - 2, nbk - 1, 1, e->stream)); // first encoder block's conv weights
            CHK(adamw_range(e, fuse_lr, 0, nbk, 2, e->stream));           // every flat item
            e->side_dirty = true;                                          // AdamW launches may still run on the side stream
            CHK(join_side(e));
            CHK(adamw_finish(e));
        } else {
            CHK(join_side(e));
            if (!e->cb) CHK(sgv_adamw_step(e, fuse_lr));
        }
        return SGV_OK;
    }
    CHK(join_side(e));
    return SGV_OK;
}
// every path out of backward_impl has joined the side stream: the main stream's position is past the last reader of the batch
static int backward_done(sgv_engine* e, int rc) { e->coll_inflight = false; if (rc == SGV_OK) x_release(e); return rc; }
int sgv_backward(sgv_engine* e, float alpha, float beta) { return backward_done(e, backward_impl(e, alpha, beta, -1.f)); }
int sgv_backward_step(sgv_engine* e, float alpha, float beta, float lr) {
    if (lr < 0.f) return fail(SGV_ERR_ARG, "negative learning rate");
    if (e && e->cb) return fail(SGV_ERR_STATE, "sgv_backward_step is the single-GPU path: with a bucket callback use sgv_backward + sgv_adamw_step_range");
    return backward_done(e, backward_impl(e, alpha, beta, lr));
}

int sgv_grad_norm(sgv_engine* e, double* out) {
    if (!e || !out) return fail(SGV_ERR_ARG, "null argument");
    CHK(lp_sync(e));
    if (opt_grad_norm(e->adam_dev, e->sn_dev, e->items_adam, e->n_items_adam, e->gnorm_part, e->stream)) return fail(SGV_ERR_HIP, "grad-norm launch failed");
    ew_rowsum_d(e->gnorm_part, e->n_items_adam, 1, e->scal + 15, 1.0, e->stream);
    double h = 0.0;
    HIPCHK(hipMemcpyAsync(&h, e->scal + 15, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *out = sqrt(h);
    return SGV_OK;
}

// which: 1 = conv-weight tiles, 2 = flat items (biases, GroupNorm affine, Linear heads), 3 = both
static int adamw_begin(sgv_engine* e) {
    if (e->adam_open) return 0;          // a bucket of this step was updated ahead of the caller's first=1 call
    e->step += 1;
    e->copies_fresh = false;
    e->wtu_fresh = false;
    e->adam_open = true;
    e->bucket_updated.assign(e->buckets.size(), 0);
    return 0;
}
// end of an optimisation step: whatever ran on the optimizer stream joins the engine stream, gradient norm^2 in a fixed order
static int adamw_finish(sgv_engine* e) {
    if (e->opt_dirty) {
        hipEvent_t ev = next_event(e);
        if (!ev) return fail(SGV_ERR_HIP, "event creation failed");
        HIPCHK(hipEventRecord(ev, e->opt));
        HIPCHK(hipStreamWaitEvent(e->stream, ev, 0));
        e->opt_dirty = false;
    }
    e->copies_fresh = true;
    e->wtu_fresh = true;
    e->adam_open = false;
    ew_rowsum_d(e->gnorm_part, e->n_items_adam_flat + e->n_items_adam_2d, 1, e->scal + 15, 1.0, e->stream);
    return 0;
}
struct AdamCoef { float b1, b2, bc1, bc2s; };
static AdamCoef adam_coef(const sgv_engine* e) {
    const double b1 = 0.9, b2 = 0.999;
    return {(float)b1, (float)b2, (float)(1.0 - pow(b1, (double)e->step)), (float)sqrt(1.0 - pow(b2, (double)e->step))};
}
static int adamw_tiles(sgv_engine* e, float lr, int t0, int t1, hipStream_t st, bool from_lp) {
    if (t1 <= t0) return 0;
    const AdamCoef c = adam_coef(e);
    // Beside the backward pass (any stream but the main one) the pass goes out in slices of 3072 64 x 64 tiles: its
    // workgroups are small and short-lived, so while one launch lasts they refill every CU the moment a slot frees, and a kernel of
    // the main stream whose workgroup needs most of a CU's LDS (the 128-row GEMM tails, the fused Conv+GroupNorm stages) is not
    // placed until the launch ends -- a kernel trace showed a 60 us tail taking 816 us beside a 1.3 ms AdamW launch.  At a launch
    // boundary the chip drains, and the waiting workgroups get their CUs.
    static const int slice_env = getenv("SGV_ADAM_SLICE") ? atoi(getenv("SGV_ADAM_SLICE")) : 3072;      // re-tuned on the final build: 2048 / 2560 / 3072 / 3584 = 11.16 / 11.11 / 11.10 / 11.11 ms
    const int slice = (st != e->stream && slice_env > 0) ? slice_env : t1 - t0;
    for (int a = t0; a < t1; a += slice) {
        const int b = std::min(t1, a + slice);
        if (opt_adamw_sn(e->adam_dev, e->sn_dev, e->items_adam_2d + a, b - a, lr, c.b1, c.b2, 1e-8f, 0.01f, c.bc1, c.bc2s, e->gnorm_part + e->n_items_adam_flat + a, e->dt, st,
                         e->grads, from_lp ? e->grads_lp : nullptr, (!from_lp && grad_lp_active(e)) ? 1 : 0))
            return fail(SGV_ERR_HIP, "adamw launch failed");
    }
    return 0;
}
static void unpack_bucket(sgv_engine* e, int b, hipStream_t st) {
    ew_unpack_bf16((const char*)e->grads_lp + 2 * e->buckets[b].first, e->grads + e->buckets[b].first, (long)e->buckets[b].second, st);
    e->bucket_packed[b] = 0;
}
// what the flat pass reads of a packed weight bucket (Linear heads)
static void unpack_bucket_flat(sgv_engine* e, int b, hipStream_t st) {
    for (auto& r : e->bucket_flat_w[b]) ew_unpack_bf16((const char*)e->grads_lp + 2 * r.first, e->grads + r.first, (long)r.second, st);
    e->bucket_packed[b] &= ~2;
}
static int adamw_range(sgv_engine* e, float lr, int bucket_lo, int bucket_hi, int which, hipStream_t st) {
    // native RCCL path: the all-reduce of every bucket touched here must have landed, and so must the small bucket's
    // (last index): it carries the <G,W> scalars every conv weight's update reads
    const int n_pend = (int)e->bucket_pending.size();
    for (int b = bucket_lo; b < n_pend; b = (b + 1 < bucket_hi ? b + 1 : (b < n_pend - 1 ? n_pend - 1 : n_pend)))
        if (e->bucket_pending[b]) {
            HIPCHK(hipStreamWaitEvent(st, e->bucket_done[b], 0));
            e->bucket_pending[b] = 0;
        }
    // the averaged bf16 wire copy: the tiled pass reads it in place (lp_direct), the flat pass gets its few weights unpacked
    const int np = (int)e->bucket_packed.size();
    for (int b = bucket_lo; b < bucket_hi && b < np; ++b) {
        if (!e->bucket_packed[b]) continue;
        if (!e->lp_direct) unpack_bucket(e, b, st);
        else if ((which & 2) && (e->bucket_packed[b] & 2)) unpack_bucket_flat(e, b, st);
    }
    // biases, GroupNorm affine and the Linear heads: flat pass.  Conv weights: tiled pass that also writes both
    // compute copies and W_new^T u for the next forward's power iteration; buckets updated ahead (adamw_bucket_async) are skipped.
    const AdamCoef c = adam_coef(e);
    const int f0 = e->flat_off[bucket_lo], f1 = e->flat_off[bucket_hi];
    if ((which & 2) && opt_adamw(e->adam_dev, e->sn_dev, e->items_adam_flat + f0, f1 - f0, lr, c.b1, c.b2, 1e-8f, 0.01f, c.bc1, c.bc2s, e->gnorm_part + f0, e->dt, st))
        return fail(SGV_ERR_HIP, "adamw launch failed");
    if (which & 1) {
        const int nu = (int)e->bucket_updated.size();
        auto skip = [&](int b) { return b < nu && e->bucket_updated[b]; };
        auto lp = [&](int b) { return b < np && (e->bucket_packed[b] & 1); };
        for (int b = bucket_lo; b < bucket_hi;) {
            if (skip(b)) { ++b; continue; }
            int h = b + 1;
            while (h < bucket_hi && !skip(h) && lp(h) == lp(b)) ++h;          // runs of buckets read from the same place
            CHK(adamw_tiles(e, lr, e->tile_off[b], e->tile_off[h], st, lp(b)));
            for (int k = b; k < h; ++k) if (k < np) e->bucket_packed[k] &= ~1;
            b = h;
        }
    }
    return 0;
}
// conv-weight AdamW of ONE weight bucket on `st`, ahead of the rest of the step: the caller has made `st` wait for the bucket's
// gradients (and their all-reduce) and for the <G,W> slots of its layers (bucket_dots)
static int adamw_bucket_async(sgv_engine* e, float lr, int b, hipStream_t st) {
    CHK(adamw_begin(e));
    const bool packed = b < (int)e->bucket_packed.size() && e->bucket_packed[b];
    if (packed && !e->lp_direct) unpack_bucket(e, b, st);
    const bool from_lp = packed && e->lp_direct && (e->bucket_packed[b] & 1);
    CHK(adamw_tiles(e, lr, e->tile_off[b], e->tile_off[b + 1], st, from_lp));
    if (from_lp) e->bucket_packed[b] &= ~1;
    e->bucket_updated[b] = 1;
    if (st != e->stream) e->opt_dirty = e->opt_dirty || st == e->opt;
    return 0;
}
int sgv_bucket_dots(const sgv_engine* e, int bucket, size_t* offset_elems, size_t* count_elems) {
    if (!e || !offset_elems || !count_elems) return fail(SGV_ERR_ARG, "null argument");
    if (bucket < 0 || bucket >= (int)e->bucket_dots.size()) return fail(SGV_ERR_ARG, "bucket %d is not a weight bucket [0,%d)", bucket, (int)e->bucket_dots.size());
    *offset_elems = e->bucket_dots[bucket].first; *count_elems = e->bucket_dots[bucket].second;
    return SGV_OK;
}
int sgv_comm_stream(sgv_engine* e, void** stream) {
    if (!e || !stream) return fail(SGV_ERR_ARG, "null argument");
    if (!ensure_comm_own(e)) return fail(SGV_ERR_HIP, "stream creation failed");
    *stream = (void*)e->comm_own;
    return SGV_OK;
}
int sgv_wire_stream(sgv_engine* e, void** stream) {
    if (!e || !stream) return fail(SGV_ERR_ARG, "null argument");
    if (!ensure_wire(e)) return fail(SGV_ERR_HIP, "the engine has no wire stream");
    *stream = (void*)e->wire;
    return SGV_OK;
}
int sgv_opt_stream(sgv_engine* e, void** stream) {
    if (!e || !stream) return fail(SGV_ERR_ARG, "null argument");
    if (!ensure_opt(e)) return fail(SGV_ERR_HIP, "the engine has no optimizer stream");
    *stream = (void*)e->opt;
    return SGV_OK;
}
int sgv_adamw_bucket_async(sgv_engine* e, float lr, int bucket) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (lr < 0.f) return fail(SGV_ERR_ARG, "negative learning rate");
    if (!ensure_opt(e)) return fail(SGV_ERR_HIP, "the engine has no optimizer stream");
    if (bucket < 0 || bucket >= (int)e->buckets.size() - 1) return fail(SGV_ERR_ARG, "bucket %d is not a weight bucket [0,%d)", bucket, (int)e->buckets.size() - 1);
    if (e->adam_open && e->bucket_updated[bucket]) return fail(SGV_ERR_STATE, "bucket %d was already updated in this step", bucket);
    return adamw_bucket_async(e, lr, bucket, e->opt);
}
int sgv_adamw_step_range(sgv_engine* e, float lr, int bucket_lo, int bucket_hi, int first, int last) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    const int nbk = (int)e->buckets.size();
    if (bucket_lo < 0 || bucket_hi > nbk || bucket_lo > bucket_hi) return fail(SGV_ERR_ARG, "bucket range [%d,%d) outside [0,%d)", bucket_lo, bucket_hi, nbk);
    if (first) CHK(adamw_begin(e));
    if (e->step < 1 || !e->adam_open) return fail(SGV_ERR_STATE, "sgv_adamw_step_range: the first call of a step must pass first=1");
    CHK(adamw_range(e, lr, bucket_lo, bucket_hi, 3, e->stream));
    if (last) CHK(adamw_finish(e));
    return SGV_OK;
}
int sgv_adamw_step(sgv_engine* e, float lr) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    const int nbk = (int)e->buckets.size();
    if (e->comm && nbk >= 3) {
        // every layer whose bucket has arrived, then the small bucket's tensors, while the last weight bucket (first
        // encoder layer, index nbk - 2) is still in flight; that layer last
        CHK(sgv_adamw_step_range(e, lr, 0, nbk - 2, 1, 0));
        CHK(sgv_adamw_step_range(e, lr, nbk - 1, nbk, 0, 0));
        return sgv_adamw_step_range(e, lr, nbk - 2, nbk - 1, 0, 1);
    }
    return sgv_adamw_step_range(e, lr, 0, nbk, 1, 1);
}
int sgv_bucket_count(const sgv_engine* e) { return e ? (int)e->buckets.size() : 0; }
int sgv_set_grad_payload(sgv_engine* e, int dtype) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (dtype != SGV_DTYPE_F32 && dtype != SGV_DTYPE_BF16) return fail(SGV_ERR_ARG, "gradient payload must be f32 or bf16");
    for (char c : e->bucket_packed) if (c) return fail(SGV_ERR_STATE, "a packed bucket is in flight: change the payload between steps");
    if (dtype == SGV_DTYPE_BF16) CHK(ensure_lp_mirror(e));
    e->payload_bf16 = dtype == SGV_DTYPE_BF16;
    e->bucket_packed.assign(e->buckets.size(), 0);
    return SGV_OK;
}
int sgv_grad_payload_buffer(sgv_engine* e, void** ptr, size_t* count) {
    if (!e || !ptr || !count) return fail(SGV_ERR_ARG, "null argument");
    if (!e->payload_bf16) return fail(SGV_ERR_STATE, "the gradient payload is the fp32 arena (sgv_grad_buffer)");
    *ptr = e->grads_lp; *count = e->n_grads;
    return SGV_OK;
}
int sgv_grad_payload_unpack(sgv_engine* e) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    for (int b = 0; b < (int)e->bucket_packed.size(); ++b)
        if (e->bucket_packed[b]) {
            if (b < (int)e->bucket_pending.size() && e->bucket_pending[b]) { HIPCHK(hipStreamWaitEvent(e->stream, e->bucket_done[b], 0)); e->bucket_pending[b] = 0; }
            ew_unpack_bf16((const char*)e->grads_lp + 2 * e->buckets[b].first, e->grads + e->buckets[b].first, (long)e->buckets[b].second, e->stream);
            e->bucket_packed[b] = 0;
        }
    return SGV_OK;
}
// device memory held by the engine, bytes: [0] fp32 master parameters, [1] gradient arena, [2] Adam m + v, [3] compute-dtype weight
// copies, [4] activations (every map of forward and backward at max_batch: nothing is recomputed), [5] split-K / reduction workspaces
int sgv_memory_info(const sgv_engine* e, size_t out[6]) {
    if (!e || !out) return fail(SGV_ERR_ARG, "null argument");
    out[0] = e->n_params * 4; out[1] = e->n_grads * 4; out[2] = e->n_grads * 8; out[3] = e->n_copies * e->esz; out[4] = e->act_bytes;
    out[5] = ((e->use_lanes ? 2 : 1) * (e->partial_floats + e->colpart_floats + e->gn_part_floats) + e->partial_tn_floats + e->red_floats + e->n_sn_tmp + e->xpose_floats) * 4;
    return SGV_OK;
}
int sgv_recompute_bytes(const sgv_engine* e, size_t* bytes) {
    if (!e || !bytes) return fail(SGV_ERR_ARG, "null argument");
    *bytes = e->recompute_bytes;
    return SGV_OK;
}
int sgv_last_grad_norm(sgv_engine* e, double* out) {
    if (!e || !out) return fail(SGV_ERR_ARG, "null argument");
    double h = 0.0;
    HIPCHK(hipMemcpyAsync(&h, e->scal + 15, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *out = sqrt(h);
    return SGV_OK;
}

// ---- per-epoch statistics without a host sync per step (reference loop: modules/train.py:171-174 reads four scalars and
// one gradient norm per parameter tensor with .item() after every step; here the step's scalars are added to a device-side
// accumulator by a one-thread kernel and read once per epoch) ----
__global__ void scalars_accumulate_kernel(const double* scal, double* acc, int n_st, double numel) {
    // acc: [0] recon (selected loss, mean), [1] kl, [2..] kl2 per stage, [8] mse, [9] gradient norm, [10] steps
    acc[0] += scal[0] / numel;
    acc[1] += scal[2];
    for (int i = 0; i + 1 < n_st; ++i) acc[2 + i] += scal[3 + i];
    acc[8] += scal[1] / numel;
    acc[9] += sqrt(scal[15]);
    acc[10] += 1.0;
}
int sgv_scalars_accumulate(sgv_engine* e) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (e->batch < 1) return fail(SGV_ERR_STATE, "no step to accumulate");
    hipLaunchKernelGGL(scalars_accumulate_kernel, dim3(1), dim3(1), 0, e->stream, e->scal, e->scal + 16, e->n_st,
                       (double)e->batch * e->T * e->N);
    return SGV_OK;
}
int sgv_scalars_read(sgv_engine* e, double* host16, int reset) {
    if (!e || !host16) return fail(SGV_ERR_ARG, "null argument");
    HIPCHK(hipMemcpyAsync(host16, e->scal + 16, 16 * 8, hipMemcpyDeviceToHost, e->stream));
    if (reset) HIPCHK(hipMemsetAsync(e->scal + 16, 0, 16 * 8, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return SGV_OK;
}

// ---- input pipeline (stateless: no engine needed) ----------------------------------------------------
int sgv_minmax_fit(const float* rows_dev, long n_rows, int n_node, float* min_dev, float* max_dev, int accumulate, void* stream) {
    if (!rows_dev || !min_dev || !max_dev) return fail(SGV_ERR_ARG, "null argument");
    if (n_rows <= 0 || n_node <= 0 || n_node % 4) return fail(SGV_ERR_ARG, "sgv_minmax_fit: n_rows > 0 and n_node %% 4 == 0 required");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SGV_ERR_NOGPU, "no HIP device visible: libsgvae has no CPU fallback");
    float* partial = nullptr;
    HIPCHK(hipMalloc((void**)&partial, sizeof(float) * 2 * (size_t)n_node * SGV_MINMAX_ROWSPLIT));
    int r = ew_minmax_fit(rows_dev, n_rows, n_node, min_dev, max_dev, partial, SGV_MINMAX_ROWSPLIT, accumulate, (hipStream_t)stream);
    hipError_t se = hipStreamSynchronize((hipStream_t)stream);
    hipFree(partial);
    if (r || se != hipSuccess) return fail(SGV_ERR_HIP, "minmax_fit failed");
    return SGV_OK;
}
int sgv_minmax_coeffs(const float* min_dev, const float* max_dev, int n_node, float lo, float hi, float* scale_dev, float* offset_dev, void* stream) {
    if (!min_dev || !max_dev || !scale_dev || !offset_dev || n_node <= 0) return fail(SGV_ERR_ARG, "bad argument");
    if (ew_minmax_coeffs(min_dev, max_dev, n_node, lo, hi, scale_dev, offset_dev, (hipStream_t)stream)) return fail(SGV_ERR_HIP, "minmax_coeffs launch failed");
    return SGV_OK;
}
int sgv_scale_convert(int dst_dtype, const float* src_dev, const float* scale_dev, const float* offset_dev, void* dst_dev, long n_rows, int n_node, void* stream) {
    if (!src_dev || !scale_dev || !offset_dev || !dst_dev) return fail(SGV_ERR_ARG, "null argument");
    if (n_node <= 0 || n_node % 8) return fail(SGV_ERR_ARG, "sgv_scale_convert: n_node %% 8 == 0 required");
    if (dst_dtype != SGV_DTYPE_F32 && dst_dtype != SGV_DTYPE_BF16) return fail(SGV_ERR_ARG, "bad dtype");
    if (ew_scale_convert(dst_dtype, src_dev, scale_dev, offset_dev, dst_dev, n_rows, n_node, (hipStream_t)stream)) return fail(SGV_ERR_HIP, "scale_convert launch failed");
    return SGV_OK;
}

size_t sgv_dataset_sample_bytes(const sgv_engine* e) { return e ? (size_t)e->N * e->T * e->esz : 0; }

int sgv_dataset_convert(sgv_engine* e, const float* src_dev, void* dst_dev, int count) {
    if (!e || !src_dev || !dst_dev) return fail(SGV_ERR_ARG, "null argument");
    ew_transpose(0, e->dt, src_dev, dst_dev, count, e->N, e->T, e->T, e->N, (long)e->N * e->T, (long)e->T * e->N, e->stream);
    return SGV_OK;
}

int sgv_augment_collate(sgv_engine* e, const void* dataset_dev, int batch, const int32_t* idx, const uint64_t* noise_seed,
                        const float* scale, const int32_t* mix_idx, const float* lam) {
    if (!e || !dataset_dev || !idx || !noise_seed || !scale || !mix_idx || !lam) return fail(SGV_ERR_ARG, "null argument");
    if (batch < 1 || batch > e->maxB) return fail(SGV_ERR_ARG, "batch %d outside [1,%d]", batch, e->maxB);
    CHK(aug_join(e));
    // small per-sample control arrays go through a device scratch at the head of xpose_tmp
    char* scratch = (char*)e->xpose_tmp;
    int* d_idx = (int*)scratch; int* d_mix = d_idx + batch;
    float* d_scale = (float*)(d_mix + batch); float* d_lam = d_scale + batch;
    unsigned long long* d_seed = (unsigned long long*)(scratch + align_up((size_t)batch * 16, 8));
    // one host->device copy of the five arrays in the device layout (pageable source: staged before the call returns)
    const size_t seed_off = align_up((size_t)batch * 16, 8), total = seed_off + (size_t)batch * 8;
    std::vector<char>& hb = e->aug_host[e->aug_turn++ & 3];           // a buffer is reused four steps later
    hb.resize(total);
    char* h = hb.data();
    memcpy(h, idx, batch * 4); memcpy(h + batch * 4, mix_idx, batch * 4);
    memcpy(h + batch * 8, scale, batch * 4); memcpy(h + batch * 12, lam, batch * 4);
    memcpy(h + seed_off, noise_seed, batch * 8);
    HIPCHK(hipMemcpyAsync(scratch, h, total, hipMemcpyHostToDevice, e->stream));
    ew_augment(e->dt, dataset_dev, e->x_in.p, (long)e->N * e->T, batch, d_idx, d_seed, d_scale, d_mix, d_lam, e->stream);
    x_release(e);
    e->batch = batch;
    e->have_fwd = false;
    return SGV_OK;
}

int sgv_augment_stage(sgv_engine* e, const void* dataset_dev, int batch, const int32_t* idx, const uint64_t* noise_seed,
                      const float* scale, const int32_t* mix_idx, const float* lam) {
    if (!e || !dataset_dev || !idx || !noise_seed || !scale || !mix_idx || !lam) return fail(SGV_ERR_ARG, "null argument");
    if (batch < 1 || batch > e->maxB) return fail(SGV_ERR_ARG, "batch %d outside [1,%d]", batch, e->maxB);
    const size_t seed_off = align_up((size_t)batch * 16, 8), total = seed_off + (size_t)batch * 8;
    if (total > AUG_CTL) return fail(SGV_ERR_ARG, "batch %d: control arrays exceed the staging scratch", batch);
    if (!ensure_aug(e)) return fail(SGV_ERR_HIP, "could not create the augmentation stream");
    // the spare buffer's last reader (the backward pass two steps back, or a pass on a batch that was staged over) has ended;
    // a batch staged before and never advanced to is replaced: same stream, so its kernels precede this copy
    const int nb = 1 - e->x_cur;
    if (e->x_free_set[nb]) HIPCHK(hipStreamWaitEvent(e->aug_stream, e->x_free[nb], 0));
    std::vector<char>& hb = e->aug_host[e->aug_turn++ & 3];           // a buffer is reused four calls later
    hb.resize(total);
    char* h = hb.data();
    memcpy(h, idx, batch * 4); memcpy(h + batch * 4, mix_idx, batch * 4);
    memcpy(h + batch * 8, scale, batch * 4); memcpy(h + batch * 12, lam, batch * 4);
    memcpy(h + seed_off, noise_seed, batch * 8);
    HIPCHK(hipMemcpyAsync(e->aug_ctl, h, total, hipMemcpyHostToDevice, e->aug_stream));
    e->aug_data = dataset_dev; e->aug_next_batch = batch;
    e->aug_staged = true; e->aug_fired = false;
    return SGV_OK;
}
int sgv_augment_advance(sgv_engine* e) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    if (!e->aug_staged) return fail(SGV_ERR_STATE, "sgv_augment_advance: no batch staged (sgv_augment_stage)");
    CHK(aug_fire(e));                        // no training forward since the stage call: the kernels go out now
    e->x_cur = 1 - e->x_cur; e->x_in = e->x_bufs[e->x_cur];
    e->batch = e->aug_next_batch;
    e->have_fwd = false;
    e->aug_staged = false; e->aug_fired = false;
    e->aug_pending = true;                   // the next reader of the batch waits for aug_done
    return SGV_OK;
}

int sgv_kernel_time_reset(sgv_engine* e, int enable) {
    if (!e) return fail(SGV_ERR_ARG, "null engine");
    hipStreamSynchronize(e->stream);
    for (auto& t : e->timers) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    e->timers.clear();
    e->timing = enable != 0;
    e->timing_detail = enable == 2;
    return SGV_OK;
}
int sgv_kernel_time_tag(sgv_engine* e, int index, char* name, size_t cap, float* total_ms, int* calls) {
    if (!e || !name || cap == 0) return fail(SGV_ERR_ARG, "null argument");
    if (index < 0 || index >= (int)e->tag_names.size()) return SGV_ERR_ARG;   // end of the list: not an error message
    HIPCHK(hipStreamSynchronize(e->stream));
    snprintf(name, cap, "%s", e->tag_names[index].c_str());
    float tot = 0.f; int n = 0;
    for (auto& t : e->timers) if (t.tag == index) { float ms = 0.f; hipEventElapsedTime(&ms, t.a, t.b); tot += ms; ++n; }
    if (total_ms) *total_ms = tot;
    if (calls) *calls = n;
    return SGV_OK;
}
int sgv_kernel_time(sgv_engine* e, const char* which, float* total_ms, int* calls) {
    if (!e || !which) return fail(SGV_ERR_ARG, "null argument");
    HIPCHK(hipStreamSynchronize(e->stream));
    auto it = e->tag_ids.find(which);
    float tot = 0.f; int n = 0;
    if (it != e->tag_ids.end()) {
        for (auto& t : e->timers) if (t.tag == it->second) { float ms = 0.f; hipEventElapsedTime(&ms, t.a, t.b); tot += ms; ++n; }
    }
    if (total_ms) *total_ms = tot;
    if (calls) *calls = n;
    return SGV_OK;
}

int sgv_test_gemm_nt(int dtype, const void* A, const void* W, void* C, const float* bias, const float* scale, const void* addend,
                     int M, int N, int K, int taps, int Tlen, int splitk, int out_f32, void* stream) {
    GemmNT p; memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.w_tap_stride = (long)N * K; p.C = C; p.ldc = N;
    p.addend = addend; p.ldadd = N; p.bias = bias; p.scale = scale;
    p.M = M; p.N = N; p.K = K; p.taps = taps; p.pad = (taps - 1) / 2; p.Tlen = Tlen; p.splitk = splitk < 1 ? 1 : splitk; p.out_f32 = out_f32;
    float* partial = nullptr;
    if (p.splitk > 1) HIPCHK(hipMalloc((void**)&partial, sizeof(float) * (size_t)p.splitk * M * N));
    p.partial = partial;
    int r = launch_gemm_nt(dtype, p, (hipStream_t)stream);
    hipError_t se = hipStreamSynchronize((hipStream_t)stream);
    if (partial) hipFree(partial);
    if (r) return fail(SGV_ERR_ARG, "launch_gemm_nt rejected the arguments (%d)", r);
    if (se != hipSuccess) return fail(SGV_ERR_HIP, "gemm_nt failed: %s", hipGetErrorString(se));
    return SGV_OK;
}

int sgv_test_gemm_nt_stats(const void* A, const void* W, void* C, const float* bias, const void* addend, int M, int N, int K,
                           int taps, int Tlen, int Cg, double* sums, void* stream) {
    if (Cg < 1 || N % Cg) return fail(SGV_ERR_ARG, "N must be a multiple of Cg");
    GemmNT p; memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.w_tap_stride = (long)N * K; p.C = C; p.ldc = N;
    p.addend = addend; p.ldadd = N; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.taps = taps; p.pad = (taps - 1) / 2; p.Tlen = Tlen; p.splitk = 1;
    p.gn_sums = sums; p.gn_Cg = Cg; p.gn_G = N / Cg;
    int r = launch_gemm_nt(SGV_DTYPE_BF16, p, (hipStream_t)stream);
    hipError_t se = hipStreamSynchronize((hipStream_t)stream);
    if (r) return fail(SGV_ERR_ARG, "launch_gemm_nt rejected the arguments (%d)", r);
    if (se != hipSuccess) return fail(SGV_ERR_HIP, "gemm_nt failed: %s", hipGetErrorString(se));
    return SGV_OK;
}

// 256x256 persistent kernel (gemm256.hip), bf16.  mode 0: forced (launch_gemm_nt256, split-K as given), 1: the engine's plan
// (gemm_nt_plan: kernel choice, split-K, main + tail rows).  sums != null: fused GroupNorm statistics (mode 0, split-K 1).
int sgv_test_conv_gn_fwd(const void* A, const void* W, const float* bias, const float* scale, const void* res, const float* gamma,
                         const float* beta, void* y, void* out, double* sums, int B, int T, int N, int K, int taps, int G, float rscale,
                         void* stream) {
    ConvGN q; memset(&q, 0, sizeof(q));
    q.A = A; q.lda = K; q.W = W; q.ldw = K; q.w_tap_stride = (long)N * K; q.bias = bias; q.scale = scale;
    q.y = y; q.ldy = N; q.out = out; q.ldout = N; q.res = res; q.ldres = N; q.rscale = rscale; q.gamma = gamma; q.beta = beta; q.sums = sums;
    q.B = B; q.T = T; q.N = N; q.K = K; q.taps = taps; q.pad = (taps - 1) / 2; q.G = G; q.Cg = G > 0 ? N / G : 0;
    if (!conv_gn_fused_eligible(SGV_DTYPE_BF16, q)) return fail(SGV_ERR_ARG, "shape not taken by the fused conv + GroupNorm kernel");
    const int r = launch_conv_gn_fwd(q, (hipStream_t)stream);
    const hipError_t se = hipStreamSynchronize((hipStream_t)stream);
    if (r || se != hipSuccess) return fail(SGV_ERR_HIP, "conv_gn launch failed (%d, %s)", r, hipGetErrorString(se));
    return SGV_OK;
}
int sgv_test_conv_gn_bwd(const void* A, const void* W, const float* scale, const void* addend, const void* premul, void* da, const void* y,
                         const double* sums, const float* gamma,
                         const float* beta, const float* cbias, void* dy, double* sums2, float* ptot, float* cdot_part, int B, int T,
                         int N, int K, int taps, int G, void* stream) {
    ConvGNBwd q; memset(&q, 0, sizeof(q));
    q.A = A; q.lda = K; q.W = W; q.ldw = K; q.w_tap_stride = (long)N * K; q.scale = scale; q.addend = addend; q.ldadd = N;
    q.premul = premul; q.ldpre = N; q.da = da; q.ldda = N;
    q.y = y; q.ldy = N; q.sums = sums; q.gamma = gamma; q.beta = beta; q.cbias = cbias; q.dy = dy; q.lddy = N;
    q.sums2 = sums2; q.ptot = ptot; q.cdot_part = cdot_part; q.rscale = 1.f; q.gscale = 1.f;
    q.B = B; q.T = T; q.N = N; q.K = K; q.taps = taps; q.pad = (taps - 1) / 2; q.G = G; q.Cg = G > 0 ? N / G : 0;
    if (!conv_gn_bwd_eligible(SGV_DTYPE_BF16, q)) return fail(SGV_ERR_ARG, "shape not taken by the fused input-gradient + GroupNorm backward kernel");
    const int r = launch_conv_gn_bwd(q, (hipStream_t)stream);
    const hipError_t se = hipStreamSynchronize((hipStream_t)stream);
    if (r || se != hipSuccess) return fail(SGV_ERR_HIP, "conv_gn_bwd launch failed (%d, %s)", r, hipGetErrorString(se));
    return SGV_OK;
}
int sgv_test_gemm_nt256(const void* A, const void* W, void* C, const float* bias, const float* scale, const void* addend, int M, int N,
                        int K, int taps, int Tlen, int splitk, int out_f32, int mode, int Cg, double* sums, int* plan_kind, void* stream) {
    GemmNT p; memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.w_tap_stride = (long)N * K; p.C = C; p.ldc = N;
    p.addend = addend; p.ldadd = N; p.bias = bias; p.scale = scale;
    p.M = M; p.N = N; p.K = K; p.taps = taps; p.pad = (taps - 1) / 2; p.Tlen = Tlen; p.splitk = splitk < 1 ? 1 : splitk; p.out_f32 = out_f32;
    const size_t cap = (size_t)32 << 20;
    float* partial = nullptr; float* part = nullptr;
    HIPCHK(hipMalloc((void**)&partial, sizeof(float) * std::max(cap, (size_t)p.splitk * M * N)));
    p.partial = partial;
    if (sums) {
        if (Cg < 1 || N % Cg) { hipFree(partial); return fail(SGV_ERR_ARG, "N must be a multiple of Cg"); }
        HIPCHK(hipMalloc((void**)&part, sizeof(float) * gemm_nt256_part_floats(M, N, 1)));
        p.gn_part = part; p.gn_sums = sums; p.gn_Cg = Cg; p.gn_G = N / Cg;
    }
    const int band_code = (mode >> 8) & 0xff, strm_code = (mode >> 16) & 7;
    p.band = band_code == 255 ? -1 : band_code;
    p.strm = strm_code == 7 ? -1 : strm_code;
    const int ts_code = (mode >> 19) & 3;
    p.ts = ts_code == 1 ? 1 : ts_code == 2 ? -1 : 0;
    const bool split_tail = (mode >> 21) & 1;          // planned launch with the 128-row tail as its own 128 x 512 launch (the engine runs it beside the main one)
    mode &= 0xff;
    int r;
    if (split_tail) {
        // the split is forced here (the planner's cost comparison and the "tail shorter than the main launch" rule decide speed, not results)
        GemmPlan pl = gemm_nt_plan(SGV_DTYPE_BF16, p, cap, 0);
        if (M % 256 != 128 || M < 384) { hipFree(partial); return fail(SGV_ERR_ARG, "split-tail test mode needs M = 128 (mod 256)"); }
        pl.kind = 2; pl.m_main = M - 128; pl.fuse_stats = 0;
        pl.sk_main = std::max(1, std::min(p.splitk, 8));
        if (plan_kind) *plan_kind = pl.kind;
        float* tp = nullptr;
        HIPCHK(hipMalloc((void**)&tp, sizeof(float) * cap));
        int sk_t = gemm_nt_tail_split(SGV_DTYPE_BF16, p, pl, cap);
        if (sk_t <= 0) {
            const long tkt = (long)taps * ((K + 63) / 64);
            sk_t = (int)std::max(1L, std::min((long)(16 / std::max(1, (N + 511) / 512)), tkt / 24));
        }
        if (N < 512) r = -4;
        else {
            r = launch_gemm_nt_main(p, pl, (hipStream_t)stream);
            if (!r) r = launch_gemm_nt_tail(p, pl, sk_t, tp, (hipStream_t)stream);
        }
        hipStreamSynchronize((hipStream_t)stream);
        hipFree(tp);
    } else if (mode == 0) {
        if (plan_kind) *plan_kind = p.ts == 1 ? 3 : 1;
        if (p.ts < 0) p.ts = 0;
        r = launch_gemm_nt256(p, (hipStream_t)stream);
    } else {
        const GemmPlan pl = gemm_nt_plan(SGV_DTYPE_BF16, p, cap, sums != nullptr);
        if (plan_kind) *plan_kind = pl.kind;
        if (sums && !pl.fuse_stats) r = -3;
        else r = launch_gemm_nt_planned(SGV_DTYPE_BF16, p, pl, (hipStream_t)stream);
    }
    hipError_t se = hipStreamSynchronize((hipStream_t)stream);
    hipFree(partial); if (part) hipFree(part);
    if (r) return fail(SGV_ERR_ARG, "the 256x256 GEMM path rejected the arguments (%d)", r);
    if (se != hipSuccess) return fail(SGV_ERR_HIP, "gemm_nt256 failed: %s", hipGetErrorString(se));
    return SGV_OK;
}

int sgv_test_stream_overlap(sgv_engine* e, int which, int* overlaps) {
    if (!e || !overlaps) return fail(SGV_ERR_ARG, "null argument");
    hipStream_t s = which == 0 ? e->lane2 : which == 1 ? e->side : which == 2 ? ensure_opt(e) : which == 3 ? ensure_comm_own(e) : nullptr;
    if (which < 0 || which > 3) return fail(SGV_ERR_ARG, "which must be 0..3");
    if (!s) { *overlaps = -1; return SGV_OK; }
    HIPCHK(hipStreamSynchronize(e->stream));
    *overlaps = streams_overlap(e->stream, s) ? 1 : 0;
    return SGV_OK;
}
// Test hook: occupy part of the chip for a bounded time (what a resident collective's channel workgroups do): `blocks` workgroups
// of `threads` threads and `lds_bytes` of LDS each spin on the constant-rate clock for `ticks` (100 MHz) on `stream`; returns at once.
__global__ void occupy_spin_kernel(long long ticks) {
    extern __shared__ char occ_lds[];
    if (threadIdx.x == 0xFFFFFF) occ_lds[0] = 1;
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
int sgv_test_occupy(void* stream, int blocks, int threads, int lds_bytes, long long ticks) {
    if (blocks < 1 || blocks > 1024 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || ticks < 0 || ticks > 1000000)
        return fail(SGV_ERR_ARG, "sgv_test_occupy: argument out of range (at most 1024 workgroups, 10 ms)");
    hipLaunchKernelGGL(occupy_spin_kernel, dim3(blocks), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, ticks);
    return hipGetLastError() == hipSuccess ? SGV_OK : fail(SGV_ERR_HIP, "occupy launch failed");
}
int sgv_test_gemm_tn(int dtype, const void* A, const void* Bm, float* dW, int M, int N1, int N2, int taps, int Tlen, int splitk,
                     int use_tr, void* stream) {
    GemmTN p; memset(&p, 0, sizeof(p));
    p.A = A; p.lda = N1; p.B = Bm; p.ldb = N2; p.out = dW; p.ldo = N2; p.out_tap_stride = (long)N1 * N2;
    p.M = M; p.N1 = N1; p.N2 = N2; p.taps = taps; p.pad = (taps - 1) / 2; p.Tlen = Tlen; p.splitk = splitk < 1 ? 1 : splitk; p.use_tr = use_tr != 0; p.force_w2 = use_tr == 2 ? 1 : use_tr == 3 ? 2 : (use_tr == 4 || use_tr == 6 || use_tr == 7) ? 3 : use_tr == 5 ? -1 : 0;
    p.out_bf16 = use_tr == 6 ? 1 : 0;          // 6: the 256 x 256 kernel with bf16 output (dW is then a bf16 array; splitk 1)
    // 7: the 256 x 256 kernel in its work-stealing form
    static int* test_sched = nullptr;          // allocated once: an allocation per call would wait for whatever else runs on the device
    if (use_tr == 7) { if (!test_sched) HIPCHK(hipMalloc((void**)&test_sched, 513 * sizeof(int))); p.sched = test_sched; }
    if (p.out_bf16 && (splitk > 1 || !gemm_tn256_eligible(dtype, p))) return fail(SGV_ERR_ARG, "sgv_test_gemm_tn: bf16 output needs the 256 x 256 kernel and splitk 1");
    float* partial = nullptr;
    const long nw = (long)taps * N1 * N2;
    if (p.splitk > 1) {
        HIPCHK(hipMalloc((void**)&partial, sizeof(float) * (size_t)p.splitk * nw));
        p.out = partial; p.out_slab_stride = nw;
    }
    int r = launch_gemm_tn(dtype, p, (hipStream_t)stream);
    if (!r && p.splitk > 1) {
        int blocks = (int)((nw + 255) / 256); if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dW, partial, p.splitk, nw);
    }
    hipError_t se = hipStreamSynchronize((hipStream_t)stream);
    if (partial) hipFree(partial);
    if (r) return fail(SGV_ERR_ARG, "launch_gemm_tn rejected the arguments (%d)", r);
    if (se != hipSuccess) return fail(SGV_ERR_HIP, "gemm_tn failed: %s", hipGetErrorString(se));
    return SGV_OK;
}

}  // extern "C"
