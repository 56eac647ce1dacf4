"""`train()` with the signature, schedules, log line, return values and output files of the reference's
modules.train.train (modules/train.py:50-256); the step itself runs in libsgvae.so.

Deviations, on purpose: scalar losses are accumulated per step but read back without forcing a
device sync per parameter tensor (the reference does ~200 `.item()` per step, train.py:156-174);
under torch.distributed the gradients are averaged with an overlapped RCCL all-reduce (the
reference's --use_ddp never synchronises gradients)."""
from __future__ import annotations

import logging
import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .VAE_network import VAE


class WarmupKLLoss:
    """train.py:18-41."""

    def __init__(self, epoch, init_beta, start_warmup, end_warmup, beta_target):
        self.epoch, self.init_beta = epoch, init_beta
        self.start_warmup, self.end_warmup, self.beta_target = start_warmup, end_warmup, beta_target

    def get_loss(self, step, losses):
        loss = 0
        for l in losses:
            loss += l
        if step < self.start_warmup:
            beta = self.init_beta
        elif self.start_warmup <= step < self.end_warmup:
            beta = (step - self.start_warmup) * (self.beta_target - self.init_beta) / (self.end_warmup - self.start_warmup) \
                + self.init_beta
        else:
            beta = self.beta_target
        return [beta, loss]


def cosine_warm_restarts_lr(base_lr, epochs, epoch, t_mult=2, eta_min_factor=1e-4):
    """LR in effect during `epoch`: CosineAnnealingWarmRestarts(T_0=epochs//4, T_mult=2, eta_min=LR*1e-4)
    stepped once per epoch (train.py:94-96,237)."""
    t0 = epochs // 4
    if t0 <= 0:
        raise ValueError(f"Expected positive integer T_0, but got {t0}")   # torch's message; epochs < 4
    eta_min = base_lr * eta_min_factor
    t_i, t_cur = t0, epoch
    while t_cur >= t_i:
        t_cur -= t_i
        t_i *= t_mult
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / t_i)) / 2


class _DevArray:
    def __init__(self, ptr, n, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def grad_payload_dtype(engine):
    """Wire format of the weight-gradient buckets: SGV_GRAD_PAYLOAD=f32|bf16, default bf16 for a bf16 engine (its gradients
    carry bf16 rounding already; the all-reduce is link-bound) and f32 for an fp32 engine."""
    v = os.environ.get("SGV_GRAD_PAYLOAD", "").lower()
    if v in ("f32", "fp32", "bf16"):
        return "bf16" if v == "bf16" else "f32"
    return "bf16" if engine.compute_dtype in ("bf16", "bfloat16") else "f32"


class _SumThenScale:
    """Work handle for backends without ReduceOp.AVG (gloo): SUM, then 1/world on the consumer's stream after the wait."""

    def __init__(self, work, seg, inv):
        self.work, self.seg, self.inv = work, seg, inv

    def wait(self):
        self.work.wait()
        self.seg.mul_(self.inv)


class _Done:
    def wait(self):
        pass


class _Both:
    """Two collectives that belong together (a weight bucket and the <G,W> scalars of its layers)."""

    def __init__(self, *works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


class GradAllReduce:
    """Bucketed mean all-reduce of the engine's flat gradient arena over RCCL, overlapped with backward (the
    engine calls `_on_bucket` as soon as the kernels producing a bucket are enqueued) and with AdamW (`step`
    updates every layer whose bucket has arrived while the last, first-encoder-layer bucket is still in flight).
    Under a gloo group (tests, debugging) the mean is SUM followed by a scale, same bucket order."""

    def __init__(self, engine, group=None, payload=None, force_collective=None):
        import weakref
        ptr, n = engine.grad_buffer()
        self._engine = weakref.ref(engine)
        self.flat = torch.as_tensor(_DevArray(ptr, n), device="cuda")
        self.group = group
        self.native_avg = dist.get_backend(group) == "nccl"
        self.inv_world = 1.0 / dist.get_world_size(group)
        # one rank: the mean over ranks is the identity, no collective is issued -- unless asked for (SGV_FORCE_COLLECTIVE=1 /
        # force_collective: the one-GPU rehearsal of the N > 1 path, every bucket goes through RCCL as it does with more ranks)
        if force_collective is None:
            force_collective = os.environ.get("SGV_FORCE_COLLECTIVE") == "1"
        self.single = dist.get_world_size(group) == 1 and not force_collective
        self.pending = []          # (bucket, work) in issue order == completion order on the RCCL stream
        self.nb = engine.bucket_count()
        # bf16 wire copy of the weight buckets (the engine packs at the fire point and unpacks in front of the bucket's AdamW)
        self.flat_lp = None
        if (payload or grad_payload_dtype(engine)) == "bf16" and not self.single:
            engine.set_grad_payload("bf16")
            lp_ptr, lp_n = engine.grad_payload_buffer()
            self.flat_lp = torch.as_tensor(_DevArray(lp_ptr, lp_n, "<i2"), device="cuda").view(torch.bfloat16)
        # optimizer overlap (include/sgvae.h: sgv_bucket_dots / sgv_adamw_bucket_async; SGV_DDP_EARLY=0 turns it off): inside
        # backward_step every weight bucket travels with the <G,W> scalars of its layers and is updated on the engine's optimizer
        # stream as soon as both collectives have landed, under the rest of backward
        self.early = not self.single and os.environ.get("SGV_DDP_EARLY", "1") != "0"
        self.lr = None
        if self.early:
            # the optimizer stream is asked for AFTER the first collective has been issued (_on_bucket): the HIP runtime hands its
            # four hardware queues out round-robin in stream-creation order, the engine's main / lane / side streams hold three
            # of them, and torch creates its NCCL stream at the first collective -- which so gets the free queue, and the
            # optimizer stream (probed by the engine against main, lane and side) the one the NCCL stream sits on: AdamW(b)
            # follows all-reduce(b) anyway.  The other order put the NCCL stream on the MAIN stream's queue.
            self.opt = None
            self.dots = [engine.bucket_dots(b) for b in range(self.nb - 1)]
            self.dots_total = sum(c for _, c in self.dots)
        # SGV_DDP_WIRE=1: released buckets are gathered (main + weight-gradient side stream) and packed on the engine's wire stream
        # and the collectives are issued from there, so backward itself never waits at a release point.  Off by default: in the
        # one-GPU rehearsal it bought nothing (the pack pass is HBM-bound wherever it runs) and one more stream shifts the
        # runtime's stream -> hardware-queue map
        self.wire = None
        if not self.single and os.environ.get("SGV_DDP_WIRE", "0") == "1":
            engine.set_option("wire_stream", 1)
            self.wire = torch.cuda.ExternalStream(engine.wire_stream())
        if not self.single:
            engine.set_bucket_callback(self._on_bucket)        # one rank: nothing to exchange, the engine keeps its single-GPU schedule

    def info(self):
        return {"ddp_path": "torch", "rccl_nranks": None, "torch_world": dist.get_world_size(self.group),
                "grad_payload": "bf16" if self.flat_lp is not None else "f32", "buckets": self.nb, "collectives_issued": not self.single}

    def backward_step(self, engine, alpha, beta, lr):
        """backward + gradient exchange + AdamW.  One rank: the engine's fused step (optimizer overlapped under backward), exactly
        the single-GPU path; several ranks: backward with bucket callbacks, then the bucket-ordered optimizer."""
        if self.single:
            engine.backward_step(alpha, beta, lr)
        else:
            self.lr = lr if self.early else None      # the callbacks update finished buckets ahead of step()
            try:
                engine.backward(alpha, beta)
            finally:
                self.lr = None
            self.step(engine, lr)

    def _mean(self, seg):
        if self.native_avg:
            return dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        return _SumThenScale(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True), seg, self.inv_world)

    def _on_bucket(self, b, off, cnt):
        if self.single:
            self.pending.append((b, _Done()))
            return
        small = b == self.nb - 1
        ahead = self.lr is not None
        if small and ahead:                                   # the conv layers' <G,W> scalars went with their buckets
            off, cnt = off + self.dots_total, cnt - self.dots_total
        seg = self.flat[off:off + cnt] if self.flat_lp is None or small else self.flat_lp[off:off + cnt]
        with torch.cuda.stream(self.wire if self.wire is not None else torch.cuda.current_stream()):
            w = self._mean(seg)
            if ahead and not small:
                doff, dcnt = self.dots[b]
                w = _Both(w, self._mean(self.flat[doff:doff + dcnt]) if dcnt else None)
        if ahead and not small:
            if b < self.nb - 2:                               # the first-encoder-layer bucket (fired last) is step()'s
                engine = self._engine()
                if self.opt is None:
                    self.opt = torch.cuda.ExternalStream(engine.opt_stream())
                with torch.cuda.stream(self.opt):
                    w.wait()                                  # stream-side: the optimizer stream waits, backward goes on
                    engine.adamw_bucket_async(self.lr, b)
                w = _Done()
        self.pending.append((b, w))

    def __call__(self, engine):
        """Wait for every bucket (use when something needs all gradients before the optimiser, e.g. sgv_grad_norm)."""
        for _, w in self.pending:
            w.wait()
        self.pending.clear()
        if self.flat_lp is not None:
            engine.grad_payload_unpack()

    def step(self, engine, lr):
        """wait(all but the final bucket) -> AdamW on those -> wait(final) -> AdamW on it.  The waits are
        stream-side (no host block), so the first AdamW call runs under the final bucket's all-reduce."""
        if not self.pending:
            engine.adamw_step(lr)
            return
        last_b, last_w = self.pending[-1]
        if len(self.pending) != self.nb or last_b != self.nb - 2:
            self(engine)                      # unexpected callback pattern: plain path
            engine.adamw_step(lr)
            return
        for _, w in self.pending[:-1]:
            w.wait()
        engine.adamw_step_range(lr, 0, last_b, True, False)
        engine.adamw_step_range(lr, last_b + 1, self.nb, False, False)
        last_w.wait()
        engine.adamw_step_range(lr, last_b, last_b + 1, False, True)
        self.pending.clear()


class _Watchdog:
    """Bounded wait around a blocking collective set-up step: if it has not been cancelled after `seconds`, print the reason and
    end the PROCESS with status 3 (a fresh process may retry; a process that has touched the GPU is never re-executed).  The
    launcher (torchrun) then takes the other ranks down, so a rank stuck inside ncclCommInitRank because a peer failed does not
    hang the job."""

    def __init__(self, seconds, what, rank):
        import threading

        def fire():
            import sys
            sys.stderr.write(f"[sgvae] rank {rank}: {what} did not finish within {seconds:.0f} s (SGV_DDP_INIT_TIMEOUT): a peer "
                             f"failed or the fabric is unreachable; exiting with status 3\n")
            sys.stderr.flush()
            os._exit(3)
        self.t = threading.Timer(seconds, fire)
        self.t.daemon = True
        self.t.start()

    def cancel(self):
        self.t.cancel()


def ddp_init_timeout():
    return float(os.environ.get("SGV_DDP_INIT_TIMEOUT", "180"))


class NativeAllReduce:
    """The same data-parallel step with the collective issued by the engine itself (include/sgvae.h: sgv_set_rccl): an
    RCCL communicator of the library's own (unique id from rank 0, broadcast over the existing torch.distributed group),
    a dedicated communication stream, bucket all-reduces launched from inside sgv_backward without a host callback, and
    sgv_adamw_step ordering the waits (first-encoder-layer bucket last).  The default on an RCCL group (make_allreduce);
    SGV_DDP_NATIVE=0 / 1 forces the torch.distributed path / this one.

    Set-up is `NativeAllReduce.create`: every step that could leave the ranks in different collectives is preceded by an
    agreement over the torch.distributed group, so either all ranks end up with a communicator or all fall back together:
      1. local probe (sgv_rccl_probe: dlopen + dlsym, no communication) -> MIN all-reduce of the flag;
      2. rank 0 draws the unique id and broadcasts a status byte WITH it (a failure on rank 0 reaches every rank);
      3. ncclCommInitRank + one 8-element mean all-reduce on the new communicator under a watchdog (_Watchdog: a rank whose
         peer failed inside the collective init cannot agree on anything any more -- it exits non-zero after the timeout);
      4. MIN all-reduce of the local result."""

    def __init__(self, engine, comm, world, group=None):
        self.engine, self.comm, self.world = engine, comm, world
        forced = os.environ.get("SGV_FORCE_COLLECTIVE") == "1"     # one-GPU rehearsal of the N > 1 path (see GradAllReduce)
        self.single = world == 1 and not forced
        self.payload = "f32"
        if not self.single and grad_payload_dtype(engine) == "bf16":
            engine.set_grad_payload("bf16")      # the engine packs, all-reduces the bf16 copy and unpacks by itself
            self.payload = "bf16"
        # the engine's own communication stream: probed so that it never shares a hardware queue with the main stream
        self.stream = torch.cuda.ExternalStream(engine.comm_stream())
        if not self.single:
            engine.set_rccl(self.comm, self.stream.cuda_stream)      # one rank: nothing to exchange, single-GPU schedule

    @classmethod
    def create(cls, engine, group=None):
        """(NativeAllReduce, None) on every rank, or (None, reason) on every rank."""
        import ctypes as C
        lib = engine.lib
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        cuda = dist.get_backend(group) == "nccl"
        dev = "cuda" if cuda else "cpu"

        def agree(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1
        # 1. can every rank resolve RCCL at all?
        local = lib.sgv_rccl_probe() == 0
        why = None if local else lib.sgv_last_error().decode()
        if not agree(local):
            return None, why or "librccl could not be resolved on another rank"
        # 2. unique id + status byte from rank 0
        msg = torch.zeros(129, dtype=torch.uint8)
        if rank == 0:
            ident = (C.c_char * 128)()
            if lib.sgv_rccl_unique_id(ident) == 0:
                msg[0] = 1
                msg[1:] = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8)
            else:
                why = lib.sgv_last_error().decode()
        msg = msg.to(dev)
        dist.broadcast(msg, src=0, group=group)
        msg = msg.cpu()
        if int(msg[0]) != 1:
            return None, why or "rank 0 could not create the RCCL unique id"
        ident = (C.c_char * 128).from_buffer_copy(bytes(msg[1:].numpy().tobytes()))
        # 3. collective init + first collective, bounded
        comm = C.c_void_p()
        dog = _Watchdog(ddp_init_timeout(), "ncclCommInitRank / the first all-reduce of the engine's communicator", rank)
        ok = lib.sgv_rccl_comm_init(C.byref(comm), world, ident, rank) == 0
        if not ok:
            why = lib.sgv_last_error().decode()
        else:
            probe = torch.full((8,), float(rank + 1), dtype=torch.float32, device="cuda")
            st = torch.cuda.current_stream()
            ok = lib.sgv_rccl_allreduce(comm, C.c_void_p(probe.data_ptr()), 8, 0, C.c_void_p(st.cuda_stream)) == 0
            if ok:
                st.synchronize()
                ok = bool(torch.allclose(probe.cpu(), torch.full((8,), (world + 1) / 2.0)))
                if not ok:
                    why = f"the first all-reduce returned {probe.cpu().tolist()[:2]}, expected {(world + 1) / 2.0}"
            else:
                why = lib.sgv_last_error().decode()
        dog.cancel()
        # 4. agree on the result
        if not agree(ok):
            if comm.value:
                lib.sgv_rccl_comm_destroy(comm)
            return None, why or "communicator set-up failed on another rank"
        return cls(engine, comm.value, world, group), None

    def info(self):
        import ctypes as C
        n = C.c_int(-1)
        self.engine.lib.sgv_rccl_comm_count(C.c_void_p(self.comm), C.byref(n))
        return {"ddp_path": "native", "rccl_nranks": int(n.value), "torch_world": self.world, "grad_payload": self.payload,
                "buckets": self.engine.bucket_count(), "collectives_issued": not self.single}

    def backward_step(self, engine, alpha, beta, lr):
        # one rank: the single-GPU schedule.  Several: sgv_backward_step on a registered communicator -- the engine averages every
        # weight bucket together with the <G,W> scalars of its layers, updates it on its optimizer stream as soon as both have
        # landed (option "ddp_early_adamw") and closes the step with sgv_adamw_step
        engine.backward_step(alpha, beta, lr)

    def step(self, engine, lr):
        engine.adamw_step(lr)           # waits for the buckets in the overlapped order (sgv_adamw_step with a communicator)

    def __call__(self, engine):
        pass                            # gradients are final once the optimiser's stream-side waits have passed

    def close(self):
        if self.comm:
            if not self.single and getattr(self.engine, "h", None):      # an engine closed before us holds nothing of ours any more
                self.engine.set_rccl(None, None)
            self.engine.lib.sgv_rccl_comm_destroy(self.comm)
            self.comm = None


def broadcast_replica_state(engine, group=None, src=0):
    """Every rank takes rank `src`'s parameters and spectral-norm u / v (all state_dict entries), tensor by tensor; returns a
    noise seed that rank `src` drew.  Replicas built from the same init seed are identical already; a state loaded from a
    file on one rank, or a different torch / numpy version on another, is not -- and nothing re-synchronises replicas later:
    the deterministic step keeps identical replicas bitwise identical, it does not repair different ones."""
    import random
    cuda = dist.get_backend(group) == "nccl"
    state = engine.state_dict()
    out = {}
    for k in sorted(state):
        t = torch.from_numpy(np.ascontiguousarray(state[k]))
        t = t.cuda() if cuda else t
        dist.broadcast(t, src=src, group=group)
        out[k] = t.cpu().numpy()
    if dist.get_rank(group) != src:
        engine.load_state(out)
    box = [random.getrandbits(48)]
    dist.broadcast_object_list(box, src=src, group=group)
    return int(box[0])


def make_allreduce(engine, group=None):
    """The data-parallel step's gradient exchange.  On an RCCL (`nccl`) group the engine issues the collectives itself
    (NativeAllReduce: its own communicator on a communication stream it placed on a hardware queue away from the main stream's,
    every bucket updated on that queue right behind its all-reduce); if the set-up fails in a way the ranks can still agree on
    (NativeAllReduce.create), every rank falls back to GradAllReduce (torch.distributed issues the bucket collectives from the
    engine's callbacks), which is also what other backends get.  SGV_DDP_NATIVE=1 / 0 forces one or the other."""
    want = os.environ.get("SGV_DDP_NATIVE", "")
    if want == "0" or (want != "1" and dist.get_backend(group) != "nccl"):
        return GradAllReduce(engine, group)
    ar, why = NativeAllReduce.create(engine, group)
    if ar is not None:
        return ar
    if want == "1":
        raise RuntimeError(f"SGV_DDP_NATIVE=1 but the engine-issued RCCL path is unavailable: {why}")
    logging.warning("engine-issued RCCL path unavailable (%s): torch.distributed issues the bucket collectives on every rank", why)
    return GradAllReduce(engine, group)


def _lookahead(it, first):
    """(current, next) pairs over `first` followed by the items of `it`; next is None for the last one."""
    cur = first
    while cur is not None:
        nxt = next(it, None)
        yield cur, nxt
        cur = nxt


def train(epochs, batch_size, train_dataloader, val_dataloader, LR, num_filter_enc, num_filter_dec, num_node, latent_dim,
          hierarchical_dim, num_time, alpha, lossfun, small, load_all, debug_mode=0, compute_dtype="bf16"):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank == 0:
        os.makedirs("checkpoints", exist_ok=True)
        os.makedirs("output", exist_ok=True)
        os.makedirs("model_save", exist_ok=True)
    # schedulers first: the reference fails here for epochs < 4 before any step runs (train.py:94-96)
    cosine_warm_restarts_lr(LR, epochs, 0)
    model = VAE(latent_dim, hierarchical_dim, num_filter_enc, num_filter_dec, num_node, num_time, lossfun=lossfun,
                batch_size=batch_size, small=small, use_checkpointing=False, compute_dtype=compute_dtype)
    eng = model._eng(batch_size)
    eng.set_option("write_xhat", 0)
    warmup_kl = WarmupKLLoss(epochs, 1e-4, int(epochs * 0.3), int(epochs * 0.8), 1)   # init_beta hard-coded
    allreduce = make_allreduce(eng) if world > 1 else None
    if world > 1:
        # identical replicas, ONE noise seed for the job: the engine keys its draws by the global sample row (sgv_set_shard), so
        # 8 x 16 samples see the noise 1 x 128 would
        eng.seed(broadcast_replica_state(eng))
        eng.set_shard(rank, world)
    if world == 1 and eng.compute_dtype in ("bf16", "bfloat16"):
        # the four big layers' weight gradients go from the GEMM to the optimizer as bf16: the rounding point
        # of the data-parallel step's bf16 wire format, which world > 1 takes by default
        eng.set_option("grad_bf16", 1)
    fused = hasattr(train_dataloader, "batch_plans")
    data = train_dataloader.resident(eng) if fused else None

    loss_print = np.zeros(epochs)
    loss_val_print = np.zeros(epochs)
    recon_print = np.zeros(epochs)
    kl_print = np.zeros(epochs)
    recon_loss_val_print = np.zeros(epochs)

    def run_forward_losses(sc, beta):
        kl = float(sum(sc["kls"]))
        recon = sc["recon"] * alpha
        return recon + kl * beta, recon, kl * beta

    for epoch in range(epochs):
        t_start = time.time()
        model.train(True)
        beta, _ = warmup_kl.get_loss(epoch, [])
        lr = cosine_warm_restarts_lr(LR, epochs, epoch)
        batches = train_dataloader.batch_plans() if fused else iter(train_dataloader)
        eng.read_accumulated(reset=True)
        if fused:
            # the DataLoader's prefetch (workers build batch i + 1 while step i runs): the engine builds the next batch beside the step
            batches = iter(batches)
            nxt = next(batches, None)
            if nxt is not None:
                eng.augment_stage(data, *nxt)
            batches = _lookahead(batches, nxt)
        for item in batches:
            if fused:
                eng.augment_advance()                   # batch i (staged during step i - 1) becomes current
                if item[1] is not None:
                    eng.augment_stage(data, *item[1])   # batch i + 1
            else:
                eng.set_input(model._prep(item))
            eng.forward(train=True, sync=False)          # nothing in the step waits for the host: the step's scalars and its
            if allreduce is not None:                   # gradient norm (train.py:156-161,171-174) are added up on the device
                allreduce.backward_step(eng, alpha, beta, lr)
            else:
                eng.backward_step(alpha, beta, lr)
            eng.accumulate_scalars()
        acc = eng.read_accumulated(reset=True)           # one read-back per epoch
        nb = acc["steps"]
        if nb == 0:
            raise ZeroDivisionError("empty training loader")
        # alpha and beta are constant within an epoch, so the epoch sums of the per-step losses follow from the summed scalars
        recon_save = acc["recon"] * alpha
        kl_save = float(sum(acc["kls"])) * beta
        loss_save = recon_save + kl_save
        grad_sum = acc["grad_norm"]
        if epoch % 20 == 0 or epoch == epochs - 1:
            model.eval()
            vl = vr = 0.0
            vb = 0
            vbatches = val_dataloader.batch_plans() if hasattr(val_dataloader, "batch_plans") else iter(val_dataloader)
            for item in vbatches:
                if hasattr(val_dataloader, "batch_plans"):
                    idx, seeds, scale, mix, lam = item
                    eng.augment_collate(val_dataloader.resident(eng), idx, seeds, scale, mix, lam)
                else:
                    eng.set_input(model._prep(item))
                sc = eng.forward(train=False)
                l, r, _ = run_forward_losses(sc, beta)
                vl += l
                vr += r
                vb += 1
            loss_val_print[epoch] = vl / vb            # ZeroDivisionError on an empty split, as the reference (D7)
            recon_loss_val_print[epoch] = vr / vb
            model.train()
        elif epoch > 0:
            loss_val_print[epoch] = loss_val_print[epoch - 1]
            recon_loss_val_print[epoch] = recon_loss_val_print[epoch - 1]
        loss_print[epoch] = loss_save / nb
        recon_print[epoch] = recon_save / nb
        kl_print[epoch] = kl_save / beta / nb
        dur = time.time() - t_start
        if rank == 0:
            logging.info("\r[Epoch {}/{}] Loss: {:.4E}   val_loss: {:.2E}   Recon:{:.4E}   Recon_val:{:.4E}   KL:{:.4E}   "
                         "Beta:{:.4E}   AvgGrad:{:.4E}   Time: {:.2f}s   ETA: {:.2f}h    LR: {:.2E}".format(
                             epoch + 1, epochs, loss_print[epoch], loss_val_print[epoch], recon_print[epoch],
                             recon_loss_val_print[epoch], kl_print[epoch], beta, grad_sum / nb, dur,
                             (epochs - epoch) * dur / 3600, lr))
    if allreduce is not None and hasattr(allreduce, "close"):
        torch.cuda.synchronize()
        allreduce.close()                 # the engine's own communicator (NativeAllReduce): unregister and destroy it
    if rank == 0:
        torch.save(model.state_dict(), "checkpoints/SimulGen-VAE.pth")
        torch.save(model, "model_save/SimulGen-VAE")
    train.last_model = model
    return loss_print, recon_print, kl_print, loss_val_print
