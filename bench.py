#!/usr/bin/env python3
"""Headline benchmark: SimulGen-VAE preset-1 `small` training step, per-GPU batch 16, on synthetic
[P x 95008 nodes x 200 timesteps] data (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = augment+collate from the HBM-resident dataset -> VAE forward (spectral-norm power
iteration, encoder, reparameterisation, decoder, recon head + loss) -> backward -> [RCCL gradient
all-reduce, N>1] -> fused AdamW.  Nothing is skipped inside the timed region.  Rank 0 prints ONE
JSON line.  Data: synthetic U(-0.7,0.7); weights: random init of the named architecture.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENC = [1024, 512, 256, 128]          # reference preset.txt:4
N_NODE, N_TIME = 95008, 200          # README.md:130-132 / BASELINE.json configs[1]
LATENT, HIER = 32, 8                 # condition.txt Latent_dim_end / Latent_dim
ALPHA, LR = 1e6, 1e-3                # condition.txt alpha / LearningR
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_BF16_PEAK_TF = 2500.0           # dense bf16 MFMA peak
MFMA_F32_PEAK_TF = 157.3


def gemm_flops(cfg, batch):
    """Algorithmic FLOPs per step of the two GEMM kernels, from the layer list (2*M*Cout*Cin*k each)."""
    from simulgen_vae_amd.spec import layer_list
    M = batch * cfg.num_time
    fwd = dx = dw = 0.0
    for l in layer_list(cfg):
        if l.op not in ("conv", "convT") or not l.used_in_forward:
            continue
        f = 2.0 * M * l.cout * l.cin * l.k
        fwd += f
        dw += f
        if not l.prefix.startswith("encoder.encoder_blocks.0.module_list.0._seq.0"):
            dx += f          # the first layer needs no input gradient
    return fwd, dx, dw


def load_traffic():
    """HBM bytes per kernel from the newest committed PMC summary (profiles/rNN_traffic.json, written by
    tools/summarize_profile.py traffic from separate FETCH_SIZE / WRITE_SIZE rocprofv3 passes of this bench)."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        t = json.load(f)
    t["_source"] = "profiles/" + os.path.basename(files[-1])
    return t


def cpu_baseline(sample_batch, cores, small=True, timed=2):
    """oracle/torch_port.py (PyTorch-CPU restatement of the reference step: same ATen conv/GroupNorm
    kernels the reference's CPU path runs) timed on this box's host cores at the stated batch (batch 16,
    full node/time/filter sizes): 1 untimed step (lazy optimizer state, allocator warm-up; its losses are the parity sample),
    then the mean of `timed` full training steps (fwd + bwd + grad-norm + AdamW)."""
    import torch
    from simulgen_vae_amd.init import init_state, synthetic_eps, synthetic_samples
    from simulgen_vae_amd.spec import VAEConfig
    from oracle.torch_port import TorchPortVAE
    torch.set_num_threads(cores)
    cfg = VAEConfig(LATENT, HIER, ENC, ENC[::-1], N_NODE, N_TIME, "MSE", small)
    m = TorchPortVAE(cfg, init_state(cfg, 7, reference_init=True))
    x = synthetic_samples(20251003, range(sample_batch), N_NODE, N_TIME)
    eps = synthetic_eps(1234, 0, cfg, sample_batch)
    r0 = m.train_step(x, eps, ALPHA, 1e-4, LR)      # untimed: lazy optimizer state, warm-up; its losses are the parity sample
    t0 = time.time()
    for _ in range(timed):
        m.train_step(x, eps, ALPHA, 1e-4, LR)
    dt = (time.time() - t0) / timed
    return sample_batch / dt, dt, r0, (x, eps)


def engine_elbo(dtype, small, x, eps):
    """ELBO (alpha*recon + beta*sum KL) of one engine forward on the CPU baseline's inputs and initial weights."""
    import torch
    from simulgen_vae_amd import engine as E
    from simulgen_vae_amd.init import init_state
    from simulgen_vae_amd.spec import VAEConfig
    cfg = VAEConfig(LATENT, HIER, ENC, ENC[::-1], N_NODE, N_TIME, "MSE", small)
    eng = E.Engine(cfg, max_batch=x.shape[0], compute_dtype=dtype)
    eng.load_state(init_state(cfg, 7, reference_init=True))
    eng.set_input(torch.from_numpy(x).cuda())
    eng.set_eps([torch.from_numpy(e).cuda() for e in eps])
    sc = eng.forward(train=True)
    eng.close()
    return ALPHA * sc["recon"] + 1e-4 * sum(sc["kls"])


def bench_latent_conditioner(args):
    """Secondary line: LatentConditionerImg training step (forward, 10*MSE+MSE, backward, clip, AdamW) on synthetic
    [B, side*side] images, preset filters 32-64-128-256-512-1024, single GPU."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    import simulgen_vae_amd  # noqa: F401
    from simulgen_vae_amd.modules.latent_conditioner_model_cnn import LatentConditionerImg
    from simulgen_vae_amd.modules.latent_conditioner import LCOptimizer
    filters = [32, 64, 128, 256, 512, 1024]          # reference preset.txt:4 (latent_conditioner_filter)
    B, side = args.batch, args.image
    m = LatentConditionerImg(filters, LATENT, (1, side, side), HIER, len(ENC) - 1, (side, side), dropout_rate=0.2, use_attention=True,
                             compute_dtype=args.dtype)
    m.train()
    opt = LCOptimizer(m, 1e-3, 1e-5)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((B, side * side), generator=g, device="cuda")
    y1 = torch.randn((B, LATENT), generator=g, device="cuda") * 0.3
    y2 = torch.randn((B, len(ENC) - 1, HIER), generator=g, device="cuda") * 0.3

    def step():
        opt.zero_grad()
        m.loss_backward(x, y1, y2)
        opt.clip_and_step(10.0, 1e-3)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    fwd_gf = {256: 304.9, 512: 1218.0}.get(side)       # SURVEY 8(f) N1, measured on the reference at batch 16
    res = {"metric": "latent-conditioner training samples/sec (preset filters, batch 16)", "value": round(B * args.steps / el, 2), "unit": "samples/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic U(0,1) images, random-init weights",
           "config": {"workload": f"LatentConditionerImg training step, {side}x{side} images, batch {B} (BASELINE.json configs[4])",
                      "filters": filters, "image": side, "per_gpu_batch": B},
           "step_tflops": round(3 * fwd_gf * (B / 16) / (el / args.steps) / 1e3, 2) if fwd_gf else None, "roofline": None, "cpu_baseline": None}
    # roofline of the dominant GEMM class: events on torch's current stream (the operators' stream) around every GEMM operator
    # call of two extra steps; algorithmic FLOPs 2*M*N*K per call (convolutions: M = output pixels, K = kh*kw*Cin)
    from simulgen_vae_amd import ops as _ops
    _ops.GEMM_TIMING = []
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    cls = {}
    if os.environ.get("SGV_LC_CALLS"):        # per-call listing of the second timed step (tools/run_lc_calls.sh)
        calls_ = _ops.GEMM_TIMING[len(_ops.GEMM_TIMING) // 2:]
        for c, fl, e0, e1 in calls_:
            ms_ = e0.elapsed_time(e1)
            print(f"[lc call] {c} {fl / 1e9:9.2f} GFLOP {ms_ * 1e3:8.1f} us {fl / ms_ / 1e9:7.0f} TFLOP/s", file=sys.stderr)
    for c, fl, e0, e1 in _ops.GEMM_TIMING:
        st = cls.setdefault(c, [0.0, 0.0, 0])
        st[0] += fl; st[1] += e0.elapsed_time(e1); st[2] += 1
    _ops.GEMM_TIMING = None
    names = {"gemm_nt": "gemm_nt_t256_kernel / gemm_nt_kernel (convolution forward and input-gradient contractions: implicit GEMMs with 2-D taps, "
                        "1x1 layers as plain GEMMs, the direct one-channel stem; sgv_op_conv2d_nt / sgv_op_gemm_nt / sgv_op_stem_conv_fwd)",
             "gemm_tn": "gemm_tn_w2_kernel / gemm_tn_kernel (weight-gradient contractions through a virtual im2col operand: up to a million rows "
                        "reduced into a small matrix, split-K slabs; sgv_op_conv2d_tn / sgv_op_gemm_tn / sgv_op_stem_conv_dw)"}
    traffic = load_traffic()
    for c, (fl, tms, calls) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
        ach = fl / (tms * 1e-3) / 1e12
        o = {"kernel": names[c], "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF,
             "unit": "TFLOP/s", "frac": round(ach / (MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF), 4), "traffic": None,
             "launches_per_step": calls // 2, "avg_launch_ms": round(tms / calls, 4), "ms_per_step": round(tms / 2, 3),
             "flop_per_launch": round(fl / calls),
             "note": "achieved = sum of 2*M*N*K over the class's operator calls / sum of their event durations (operator = main kernel + its "
                     "split-K combine); traffic = (FETCH_SIZE x2 + WRITE_SIZE) per launch from the committed rocprofv3 --pmc passes of this command"}
        t = (traffic or {}).get("lc_" + c)
        if t and t.get("launches"):
            o["traffic"] = round((t["fetch_bytes"] + t["write_bytes"]) / t["launches"])
            o["traffic_source"] = traffic.get("_source")
        if res["roofline"] is None:
            res["roofline"] = o
        else:
            res["roofline_" + c] = o
    if args.cpu_baseline == "auto":
        # the CPU restatement (oracle/lc_torch_port.py, PyTorch-CPU fp32) on the host cores: 1 untimed + 1 timed training
        # step at the same image size, batch 4
        from oracle.lc_torch_port import TorchPortLC
        cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), args.cpu_threads)
        torch.set_num_threads(cores)
        cb = 4
        port = TorchPortLC(filters, LATENT, HIER, len(ENC) - 1, {k: v.clone() for k, v in m.state_dict().items()}, dropout_rate=0.2)
        xc, y1c, y2c = x[:cb].cpu(), y1[:cb].cpu(), y2[:cb].cpu()
        port.loss_backward(xc, y1c, y2c)
        port.clip_and_step(1e-3, 1e-5)
        t0 = time.perf_counter()
        port.loss_backward(xc, y1c, y2c)
        port.clip_and_step(1e-3, 1e-5)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(cb / dt, 3), "unit": "samples/s", "cores": cores, "kind": "port",
                               "sample": f"oracle/lc_torch_port.py, 1 timed training step at {side}x{side}, batch {cb} ({dt:.1f}s)"}
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--size", default="small", choices=["small", "large"],
                    help="small = the headline config (BASELINE.json configs[1]); large = configs[3] (secondary; all "
                         "activations stay resident in the 288 GB HBM, so no recompute is needed)")
    ap.add_argument("--dataset", type=int, default=484, help="synthetic samples resident in HBM per rank")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "skip"])
    ap.add_argument("--cpu-sample-batch", type=int, default=16, help="batch of the timed CPU steps (BASELINE.md section 3 states 16)")
    ap.add_argument("--cpu-timed-steps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true", help="build every batch at the start of its own step (sgv_augment_collate) instead of beside the previous step (sgv_augment_stage / sgv_augment_advance, the training loop's way)")
    ap.add_argument("--host-times", action="store_true", help="diagnostic: print the host's time inside each engine call")
    ap.add_argument("--recompute", action="store_true",
                    help="measurement only: regenerate the GroupNorm + GELU outputs in backward (what configs[3]'s '+ grad-checkpoint' would cost); "
                         "adds config.recompute_activations / recompute_gib to the line")
    ap.add_argument("--layer-times", action="store_true", help="print a per-layer GEMM table (stderr) after the run")
    ap.add_argument("--workload", default="vae", choices=["vae", "lc"],
                    help="vae = the headline hot path; lc = image latent-conditioner training step (BASELINE.json configs[4], secondary)")
    ap.add_argument("--image", type=int, default=512, help="lc: image side (configs[4] names 512x512)")
    args = ap.parse_args()
    if args.workload == "lc":
        return bench_latent_conditioner(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs: there is no CPU fallback for the product path")
    torch.cuda.set_device(local)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    import simulgen_vae_amd  # noqa: F401
    from simulgen_vae_amd import engine as E
    from simulgen_vae_amd.init import init_state
    from simulgen_vae_amd.spec import VAEConfig
    from simulgen_vae_amd.modules.train import make_allreduce

    cfg = VAEConfig(LATENT, HIER, ENC, ENC[::-1], N_NODE, N_TIME, "MSE", args.size == "small")
    B = args.batch
    eng = E.Engine(cfg, max_batch=B, compute_dtype=args.dtype)
    t_init = time.time()
    eng.load_state(init_state(cfg, 7, reference_init=True))   # same seed on every rank -> identical replicas
    eng.set_option("write_xhat", 0)   # train.py:142 discards the reconstruction (`_`)
    if args.recompute:
        eng.set_option("recompute_activations", 1)
    eng.seed(1234)                     # ONE noise seed for the job; the draws are keyed by the global sample row (sgv_set_shard)
    eng.set_shard(rank, world)
    # synthetic dataset, generated on the device, converted to the engine's resident layout (replicated: same samples on every rank)
    P = args.dataset
    gen = torch.Generator(device="cuda").manual_seed(20251003)
    esz = 2 if args.dtype == "bf16" else 4
    data = torch.empty(P * N_NODE * N_TIME * esz, dtype=torch.uint8, device="cuda")
    chunk = 8
    for p0 in range(0, P, chunk):
        c = min(chunk, P - p0)
        src = (torch.rand((c, N_NODE, N_TIME), generator=gen, device="cuda", dtype=torch.float32) * 1.4 - 0.7)
        eng.dataset_convert(src, data[p0 * N_NODE * N_TIME * esz:], c)
    torch.cuda.synchronize()
    del src
    if rank == 0:
        print(f"[bench] init {time.time() - t_init:.1f}s, dataset {P} samples resident "
              f"({data.numel() / 1e9:.1f} GB), dtype {args.dtype}", file=sys.stderr)

    ddp = world > 1 or (os.environ.get("SGV_FORCE_DDP") == "1" and dist.is_initialized())   # forced: plumbing test at N=1
    grad_bf16 = not ddp and args.dtype == "bf16" and os.environ.get("SGV_BENCH_GRAD_BF16", "1") != "0"
    if grad_bf16:
        eng.set_option("grad_bf16", 1)    # as modules/train.py does on one GPU: the big layers' gradients reach AdamW as bf16 (the data-parallel step's wire rounding)
    # bucketed mean all-reduce over RCCL, overlapped with backward and with AdamW (modules/train.py GradAllReduce)
    allreduce = make_allreduce(eng) if ddp else None
    ddp_info = allreduce.info() if allreduce is not None else {"ddp_path": None, "rccl_nranks": None, "torch_world": world,
                                                                "grad_payload": None, "buckets": eng.bucket_count(), "collectives_issued": False}
    # which of the engine's auxiliary streams really run beside the main stream (hardware-queue probe): the lane
    # and the weight-gradient side stream always exist; the optimizer / communication streams only on the data-parallel path
    if ddp:
        ddp_info["aux_streams_overlap"] = eng.stream_overlaps()
    else:
        so = eng.stream_overlaps_existing()
        ddp_info["aux_streams_overlap"] = so

    rng = random.Random(99)              # the same stream on every rank: the GLOBAL batch is drawn, rank r keeps rows r::world
    nprng = np.random.RandomState(5)
    epochs_beta = 1e-4   # WarmupKLLoss initial beta (train.py:75-81)

    def draw_plan():
        # AugmentedDataset decisions (augmentation.py:58-84) drawn on the host, applied on the device
        idx, seeds, scale, mix, lam = [], [], [], [], []
        for g in range(B * world):
            i_ = rng.randrange(P)
            sd_ = rng.getrandbits(63) | 1 if rng.random() < 0.5 else 0
            sc_ = 0.9 + rng.random() * 0.2 if rng.random() < 0.5 else 1.0
            if rng.random() < 0.5 and P > 1:
                o = rng.randrange(P)
                while o == i_:
                    o = rng.randrange(P)
                m_, l_ = o, max(0.1, min(float(nprng.beta(0.2, 0.2)), 0.9))
            else:
                m_, l_ = -1, 1.0
            if g % world == rank:
                idx.append(i_); seeds.append(sd_); scale.append(sc_); mix.append(m_); lam.append(l_)
        return idx, seeds, scale, mix, lam

    prefetch = not args.no_prefetch and os.environ.get("SGV_BENCH_NO_PREFETCH") != "1"     # the env form is for A/B scripts
    if prefetch:
        eng.augment_stage(data, *draw_plan())          # modules/train.py's loop: batch i + 1 is built beside step i

    def one_step(step_idx):
        if prefetch:
            eng.augment_advance()
            eng.augment_stage(data, *draw_plan())
        else:
            eng.augment_collate(data, *draw_plan())
        eng.forward(train=True, sync=False)
        if ddp:
            allreduce.backward_step(eng, ALPHA, epochs_beta, LR)      # one rank: the fused single-GPU step, no collective
        else:
            eng.backward_step(ALPHA, epochs_beta, LR)     # backward + AdamW, optimizer overlapped under backward

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if args.host_times and rank == 0:
        # diagnostic: the host's time inside each engine call of three back-to-back steps that start on an idle GPU (no
        # back-pressure from a full queue): is the step enqueued faster than the GPU runs it?
        def timed(f, *a, **k):
            t = time.perf_counter(); f(*a, **k); return (time.perf_counter() - t) * 1e3
        idx = list(range(B)); z = [0] * B; o = [1.0] * B; m1 = [-1] * B
        for rep in range(3):
            ta = timed(eng.augment_collate, data, idx, z, o, m1, o)
            tf = timed(eng.forward, train=True, sync=False)
            tb = timed(eng.backward_step, ALPHA, epochs_beta, LR)
            print(f"[bench] host ms: augment_collate {ta:.3f}  forward {tf:.3f}  backward_step {tb:.3f}", file=sys.stderr)
        t = time.perf_counter(); torch.cuda.synchronize()
        print(f"[bench] drain after the three steps: {(time.perf_counter() - t) * 1e3:.3f} ms", file=sys.stderr)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    sc = eng.forward(train=False)   # sanity: losses are finite after the timed steps
    finite = bool(np.isfinite(sc["recon"]) and all(np.isfinite(k) for k in sc["kls"]))

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        fwd, dx, dw = gemm_flops(cfg, B)
        peak = MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF
        roof = {"kernel": "gemm_nt_t256_kernel + gemm_nt_wide64p_kernel + gemm_nt_kernel (conv forward + input-gradient implicit GEMMs)", "bound": "mfma",
                "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "flop_per_step": fwd + dx}
        result = {"metric": f"simulation samples/sec/node (preset-1 {args.size}, batch {B})", "value": round(value, 3),
                  "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                  "dtype": args.dtype, "data": "synthetic U(-0.7,0.7) [P x 95008 x 200], random-init weights",
                  "config": {"workload": f"preset=1 --size={args.size}, synthetic [P x 200 x 95008], batch {B} per GPU "
                                         f"(BASELINE.json configs[{1 if args.size == 'small' else 3}])", "per_gpu_batch": B, "global_batch": B * world,
                             "num_node": N_NODE, "num_time": N_TIME, "filters": ENC, "dataset_samples_per_gpu": P,
                             "parallelism": f"dp{world}", "losses_finite": finite},
                  "step_tflops": round((fwd + dx + dw) / (ms * 1e-3) / 1e12, 2),
                  "roofline": roof}
        result["config"].update(ddp_info)
        result["config"]["weight_grad_storage"] = "bf16 for the 256x256-TN layers (engine option grad_bf16)" if grad_bf16 else ("bf16 wire copy" if ddp_info.get("grad_payload") == "bf16" else "f32")
        result["config"]["prefetched_augmentation"] = bool(prefetch)
        mem = eng.memory_info()
        result["config"]["resident_gib"] = round(sum(mem.values()) / 2 ** 30, 2)
        if args.recompute:
            result["config"]["recompute_activations"] = True
            result["config"]["recompute_gib"] = round(eng.recompute_bytes() / 2 ** 30, 3)
    # per-kernel durations (hipEvents on the engine's stream around each GEMM's main kernel), outside the timed region
    if not args.no_kernel_timing:
        eng.kernel_time_reset(2)
        nt = 2
        for i in range(nt):
            one_step(10_000 + i)
        tags = eng.kernel_time_tags()
        eng.kernel_time_reset(False)
        # the optimizer pass alone (in the step it is spread under backward): lr 0 leaves the weights, the moments take a stale update
        adamw_ms = None
        if rank == 0:
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            eng.adamw_step(0.0)
            evs[0].record(); eng.adamw_step(0.0); evs[1].record()
            torch.cuda.synchronize()
            adamw_ms = evs[0].elapsed_time(evs[1])
        if rank == 0:
            cls_stat = {}
            enc = [0.0, 0.0, 0]      # north_star: "the encoder conv stack" = every convolution under encoder.* (forward, dX, dW)
            for tag, tms, calls in tags:
                if "|" not in tag:
                    continue
                cls, _layer, shape = tag.split("|")
                d = dict(kv.split("=") for kv in shape.split())
                fl = 2.0 * int(d["M"]) * int(d["N"]) * int(d["K"]) * int(d["taps"]) * calls
                st = cls_stat.setdefault(cls, [0.0, 0.0, 0])
                st[0] += fl; st[1] += tms; st[2] += calls
                if _layer.startswith("encoder."):
                    enc[0] += fl; enc[1] += tms; enc[2] += calls
            if enc[2] and enc[1] > 0:
                ach = enc[0] / (enc[1] * 1e-3) / 1e12
                result["encoder_conv_stack"] = {
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "launches_per_step": enc[2] // nt, "ms_per_step": round(enc[1] / nt, 3), "flop_per_step": round(enc[0] / nt),
                    "note": "all convolution launches of encoder.* (forward, input gradient, weight gradient; the fused Conv+GroupNorm+GELU "
                            "stage kernels are counted with their whole duration): sum of 2*M*N*K*taps / sum of hipEvent durations"}
            traffic = load_traffic()

            def roof_obj(cls, kernel, note):
                fl, tms, calls = cls_stat.get(cls, (0.0, 0.0, 0))
                if calls == 0 or tms <= 0:
                    return None
                ach = fl / (tms * 1e-3) / 1e12
                o = {"kernel": kernel, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(ach / peak, 4), "traffic": None, "launches_per_step": calls // nt,
                     "avg_launch_ms": round(tms / calls, 4), "ms_per_step": round(tms / nt, 3),
                     "flop_per_launch": round(fl / calls), "note": note}
                t = traffic.get(kernel.split(" ")[0]) if traffic else None
                if t and t.get("launches"):
                    o["traffic"] = round((t["fetch_bytes"] + t["write_bytes"]) / t["launches"])
                    o["traffic_source"] = traffic.get("_source")
                return o
            acct = ("achieved = sum over its launches of 2*M*N*K*taps / sum of hipEvent durations of the launch groups on the main stream (main "
                    "kernel + its split-K combine + the 128-row tail of M = 3200: a second launch behind the main one, or -- K = 95008 and "
                    "2560^2 x 5 -- a 16-workgroup launch of the 128x512 tile shape BESIDE it on the lane stream, joined before the second event); "
                    "traffic = (FETCH_SIZE x2 + WRITE_SIZE) per launch from the committed rocprofv3 --pmc passes")
            result["roofline"] = roof_obj(
                "gemm_nt_t256", "gemm_nt_t256_kernel (conv forward / input-gradient implicit GEMM: 256x256 tiles, persistent, 8 waves, "
                "quarter-refilled LDS-DMA double buffer)", acct) or result["roofline"]
            result["roofline_gemm_nt_wide"] = roof_obj(
                "gemm_nt_wide", "gemm_nt_wide64p_kernel (128x256 tiles, LDS-DMA ring: mid-size layers and the 128-row tails)", "same accounting")
            result["roofline_gemm_nt_128"] = roof_obj("gemm_nt", "gemm_nt_kernel (128x128 tiles: N < 256 or short K)", "same accounting")
            if adamw_ms:
                n_par = sum(int(np.prod(shape)) for _n, shape, _k, hg in eng.param_info() if hg)
                ab = 28.0 * n_par              # w, m, v read + written in fp32 (24 B) + the fp32 gradient read (4 B); VERDICT r2 #7's accounting
                o = {"kernel": "adamw_sn_kernel (AdamW over every trainable tensor + the bf16 operand copies + the spectral-norm <W,v> partials, one pass)",
                     "bound": "hbm", "achieved": round(ab / (adamw_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ab / (adamw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "params": n_par,
                     "algorithmic_bytes": round(ab), "ms_alone": round(adamw_ms, 3),
                     "note": "timed alone on the main stream after the timed region (torch events on the engine's stream, lr 0); in the step "
                             "it runs bucket by bucket on the side stream under backward.  The pass also writes the two bf16 operand copies "
                             "(+4 B/param for conv weights), which the 28 B/param do not count"}
                t = traffic.get("adamw_sn_kernel") if traffic else None
                if t:
                    o["traffic"] = round(t["fetch_bytes"] + t["write_bytes"])      # per step: the launches of one step together are one pass
                    o["traffic_source"] = traffic.get("_source")
                result["roofline_adamw"] = o
            result["roofline_gemm_tn"] = roof_obj("gemm_tn", "gemm_tn_t256_kernel + gemm_tn_w2_kernel (weight-gradient GEMMs: the four big ones on the persistent 256x256 kernel with transposed LDS reads, the rest on 128x256 tiles with two blocks per CU; layers with fewer than 256 input channels on the 128x128 gemm_tn_kernel)", "same accounting; traffic sums the kernels of the class")
    if args.layer_times and rank == 0 and world == 1:      # extra step on one rank only: never with collectives in the step
        eng.kernel_time_reset(2)
        one_step(20_000)
        rows = []
        for tag, tms, calls in eng.kernel_time_tags():
            if "|" not in tag:
                continue
            cls, layer, shape = tag.split("|")
            d = dict(kv.split("=") for kv in shape.split())
            fl = 2.0 * int(d["M"]) * int(d["N"]) * int(d["K"]) * int(d["taps"])
            rows.append((tms / calls, cls, layer, shape, fl / (tms / calls * 1e-3) / 1e12))
        eng.kernel_time_reset(False)
        print("[bench] per-layer GEMM times (one step, hipEvents around the main kernel of each launch):", file=sys.stderr)
        for tms, cls, layer, shape, tf in sorted(rows, reverse=True):
            print(f"  {tms * 1e3:8.1f} us  {tf:7.1f} TF/s  {cls:8s} {layer:58s} {shape}", file=sys.stderr)
    if rank == 0:
        if world == 1 and args.cpu_baseline == "auto":
            eng.close()
            del data
            torch.cuda.empty_cache()
            cores = os.cpu_count() or 1
            try:
                cores = len(os.sched_getaffinity(0))
            except Exception:
                pass
            cores = min(cores, args.cpu_threads)   # a 1-GPU box's CPU share is 16 cores; 256 torch threads thrash
            v, dt, r, (cx, ceps) = cpu_baseline(args.cpu_sample_batch, cores, args.size == "small", args.cpu_timed_steps)
            # parity beside the timing: the engine's ELBO on the very batch / weights the CPU port just ran (first step)
            e_elbo = engine_elbo(args.dtype, args.size == "small", cx, ceps)
            result["elbo_rel_vs_cpu_port"] = float(abs(e_elbo - r["loss"]) / abs(r["loss"]))
            result["cpu_baseline"] = {"value": round(v, 4), "unit": "samples/s", "cores": cores, "kind": "port",
                                      "sample": f"oracle/torch_port.py (PyTorch-CPU fp32 port of the reference step), 1 warm-up + mean of "
                                                f"{args.cpu_timed_steps} timed full training steps (fwd+bwd+grad-norm+AdamW) at full "
                                                f"N=95008/T=200/filters with batch {args.cpu_sample_batch} ({dt:.1f} s per step); the "
                                                f"reference itself measured in the survey container: 0.564 samples/s on 8 cores at batch 16"}
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if allreduce is not None and hasattr(allreduce, "close"):
        torch.cuda.synchronize()
        try:
            allreduce.close()             # the engine's own RCCL communicator (rank 0 with --cpu-baseline has closed its engine already)
        except Exception:                 # noqa: BLE001
            pass
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
