"""The reference-shaped Python surface (modules.VAE_network.VAE, modules.train.train) on the GPU."""
import os
import random

import numpy as np
import pytest
import torch

import simulgen_vae_amd
from tests.gpu_common import G0, GOLD, make_cfg, relerr
from simulgen_vae_amd.init import synthetic_eps, synthetic_samples
from simulgen_vae_amd.spec import param_spec

simulgen_vae_amd.install_reference_api()
from modules.VAE_network import VAE  # noqa: E402
from modules import augmentation as aug  # noqa: E402
from modules import train as tr  # noqa: E402

pytestmark = pytest.mark.gpu


def test_vae_api_matches_reference_golden():
    """VAE.encoder / VAE.decoder(mode='fix') / state_dict round trip on the reference's step-3 state."""
    g = np.load(os.path.join(GOLD, "g0_small_MSE.npz"))
    cfg = make_cfg(G0)
    B = int(g["meta"][6])
    m = VAE(cfg.latent_dim, cfg.hierarchical_dim, cfg.num_filter_enc, cfg.num_filter_dec, cfg.num_node, cfg.num_time,
            lossfun="MSE", batch_size=B, small=True, compute_dtype="f32")
    sd = {e.name: torch.from_numpy(g["s3." + e.name]) for e in param_spec(cfg)}
    m.load_state_dict(sd)
    m.eval()
    x = torch.from_numpy(synthetic_samples(int(g["meta"][4]), range(100, 100 + B), cfg.num_node, cfg.num_time))
    mu, lv, xs = m.encoder(x)
    assert relerr(mu.cpu().numpy(), g["eval.mu"]) < 3e-4 and relerr(lv.cpu().numpy(), g["eval.log_var"]) < 3e-4
    assert len(xs) == 3 and relerr(xs[0].cpu().numpy(), g["eval.xs0"]) < 3e-4
    eps = synthetic_eps(int(g["meta"][5]), 100, cfg, B)
    m._eng().set_eps([torch.from_numpy(eps[0]).cuda()] + [torch.from_numpy(e).cuda() for e in eps[1:]])
    xf, kls = m.decoder(torch.from_numpy(g["fix.z"]), xs, mode="fix")
    assert relerr(xf.cpu().numpy(), g["fix.x_hat"]) < 3e-4 and len(kls) == 2
    out = m.state_dict()
    assert list(out.keys()) == [e.name for e in param_spec(cfg)]
    for k in ("decoder.recon.0.weight_orig", "decoder.decoder_blocks.1.module_list.0._seq.0.weight_orig",
              "encoder.xs_linear.2.weight_v", "decoder.sequence_start.0.0.bias"):
        np.testing.assert_array_equal(out[k].numpy(), g["s3." + k])
    x_hat, recon, kl_list, mse = m(x)
    assert x_hat.shape == (B, cfg.num_node, cfg.num_time) and len(kl_list) == 3
    assert abs(float(recon) - float(mse)) < 1e-9


def test_train_plumbing_run(tmp_path, monkeypatch):
    """BASELINE.json configs[0]-style plumbing through modules.train.train on synthetic data (8 params,
    4 epochs: the smallest the reference itself accepts), then reload the saved model."""
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    random.seed(0)
    np.random.seed(0)
    enc = [64, 32, 16, 8]
    N, T, P, B, E = 512, 16, 8, 4, 4
    x = synthetic_samples(20251003, range(P), N, T)
    tl, vl = aug.create_augmented_dataloaders(x, B, load_all=True)
    out = tr.train(E, B, tl, vl, 1e-3, enc, enc[::-1], N, 32, 8, T, 1e6, "MSE", True, True, compute_dtype="f32")
    loss, recon, kl, val = out
    assert all(len(a) == E for a in out) and np.isfinite(np.concatenate(out)).all()
    assert val[0] > 0 and val[-1] > 0 and val[1] == val[0]           # validation at epoch 0 and the last only
    assert loss[-1] < loss[0]
    assert os.path.exists("checkpoints/SimulGen-VAE.pth") and os.path.exists("model_save/SimulGen-VAE")
    sd = torch.load("checkpoints/SimulGen-VAE.pth", weights_only=True)
    assert len(sd) == 240
    m2 = torch.load("model_save/SimulGen-VAE", weights_only=False)    # our own file (pickled mirror object)
    m2.eval()
    m1 = tr.train.last_model
    m1.eval()
    xb = torch.from_numpy(x[:B])
    mu1 = m1.encoder(xb)[0].cpu().numpy()
    mu2 = m2.encoder(xb)[0].cpu().numpy()
    np.testing.assert_allclose(mu1, mu2, rtol=1e-4, atol=1e-5)   # split-K heads use float atomics: order varies
    with pytest.raises(ValueError):
        tr.train(2, B, tl, vl, 1e-3, enc, enc[::-1], N, 32, 8, T, 1e6, "MSE", True, True)   # epochs < 4


def test_evaluate_vae_reconstruction_matches_oracle(tmp_path, monkeypatch):
    """evaluate_vae_reconstruction / export_latents against the CPU oracle (eval-mode encoder ->
    z = mu (+ injected eps) -> mode='fix' decoder -> per-batch MSE), including the reference's first-sample-of-
    each-batch bookkeeping and the on-disk formats of model_save/latent_vectors.npy, xs.npy and the L2 loss file."""
    from modules import utils as U
    from oracle.vae_oracle import OracleVAE
    monkeypatch.chdir(tmp_path)
    g = np.load(os.path.join(GOLD, "g0_small_MSE.npz"))
    cfg = make_cfg(G0)
    state = {e.name: g["s3." + e.name] for e in param_spec(cfg)}
    P = 6
    x = synthetic_samples(77, range(P), cfg.num_node, cfg.num_time)
    m = VAE(cfg.latent_dim, cfg.hierarchical_dim, cfg.num_filter_enc, cfg.num_filter_dec, cfg.num_node, cfg.num_time,
            lossfun="MSE", batch_size=2, small=True, compute_dtype="f32")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    m.eval()
    rng = np.random.RandomState(3)
    eps_tab = rng.standard_normal((P, 2, cfg.latent_dim)).astype(np.float32)

    def eps_fn(j, i, like):
        return torch.from_numpy(np.repeat(eps_tab[j, i][None], like.shape[0], 0)).to(like.device)

    orc = OracleVAE(cfg, state)
    orc.training = False
    orc._W, orc._sigma, orc.acts = {}, {}, {}     # eval mode: effective weights are constant, cache them once
    zero_maps = [np.zeros((1, c, cfg.num_time), np.float32) for c in cfg.num_filter_dec[1:-1]]

    def oracle_row(sample, eps):
        mu, lv, xs = orc.encoder(sample[None])
        z = mu + eps[None] * np.clip(np.exp(0.5 * lv), 1e-8, 10.0)
        xh, _ = orc.decoder(z.astype(np.float32), xs, zero_maps, mode="fix")
        return z[0], [v[0] for v in xs], xh[0], float(np.mean((xh[0] - sample) ** 2))

    # batch-2 loader, two draws per batch: row j describes the FIRST sample of batch j, best of the two draws
    loader = torch.utils.data.DataLoader(U.Dataset(x, False), batch_size=2, shuffle=False)
    lat, hier, rl, rec, tot = U.evaluate_vae_reconstruction(m, loader, "cuda", P, cfg.num_filter_enc, cfg.hierarchical_dim,
                                                            cfg.latent_dim, recon_iter=2, dataset_name="Unit (test)",
                                                            save_images=False, eps_fn=eps_fn)
    assert lat.shape == (P, cfg.latent_dim) and hier.shape == (P, 3, cfg.hierarchical_dim) and rec.shape == (P, cfg.num_node, cfg.num_time)
    assert np.all(lat[3:] == 0) and np.all(rl[3:] == 0)            # only len(loader) rows are ever written
    for j in range(3):
        mu, lv, xs = orc.encoder(x[2 * j:2 * j + 2])
        best = None
        for i in range(2):
            z = (mu + eps_tab[j, i][None] * np.clip(np.exp(0.5 * lv), 1e-8, 10.0)).astype(np.float32)
            xh, _ = orc.decoder(z, xs, [np.zeros((2, c, cfg.num_time), np.float32) for c in cfg.num_filter_dec[1:-1]], mode="fix")
            mse = float(np.mean((xh - x[2 * j:2 * j + 2]) ** 2))
            if best is None or mse < best[0]:
                best = (mse, z[0], xh[0])
        assert abs(rl[j] - best[0]) < 2e-4 * best[0]
        assert relerr(lat[j], best[1]) < 3e-4 and relerr(rec[j], best[2]) < 3e-4
        for k in range(3):
            assert relerr(hier[j, k], xs[k][0]) < 3e-4
    # whole-dataset export (batch 1): files in the reference's formats
    lat1, hier1, rl1 = U.export_latents(m, x, cfg.num_filter_enc, cfg.hierarchical_dim, cfg.latent_dim, recon_iter=1,
                                        eps_fn=lambda j, i, like: torch.zeros_like(like))
    z0, xs0, xh0, mse0 = oracle_row(x[4], np.zeros(cfg.latent_dim, np.float32))
    assert relerr(lat1[4], z0) < 3e-4 and abs(rl1[4] - mse0) < 2e-4 * mse0
    a = np.load("model_save/latent_vectors.npy")
    b = np.load("model_save/xs.npy")
    c = np.loadtxt("SimulGen-VAE_L2_loss.txt")
    assert a.dtype == np.float64 and a.shape == (P, cfg.latent_dim) and b.shape == (P, 3, cfg.hierarchical_dim)
    np.testing.assert_allclose(c, rl1, rtol=1e-6)
    assert abs(U.evaluate_vae_simple(m, loader, "cuda", "Unit", eps_fn=eps_fn) - tot) < 1e-6 + 0.5 * tot   # same loop, last draw vs best


def test_overlapped_allreduce_step_equals_plain_step():
    """the data-parallel step (bucket callbacks -> RCCL mean all-reduce -> bucket-ranged AdamW with the
    first-encoder-layer bucket updated last) must leave exactly the state of the plain single-GPU step.  One-rank
    RCCL group: AVG over one rank is the identity, so every parameter, u/v vector and Adam moment has to match
    (up to float-atomic ordering); the bucket order contract (small bucket released before the last weight bucket) is checked too."""
    import torch.distributed as dist
    from modules.train import GradAllReduce
    from simulgen_vae_amd.engine import Engine
    from simulgen_vae_amd.init import init_state
    from tests.gpu_common import G1
    cfg = make_cfg(G1)
    B = 4
    x = torch.from_numpy(synthetic_samples(5, range(B), cfg.num_node, cfg.num_time)).cuda()
    state = init_state(cfg, 11, reference_init=True)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    try:
        outs = []
        for mode in ("plain", "ddp"):
            eng = Engine(cfg, max_batch=B, compute_dtype="f32")
            eng.load_state(state)
            eng.seed(99)
            ar = GradAllReduce(eng) if mode == "ddp" else None
            order = []
            if ar is not None:
                inner = ar._on_bucket
                ar._on_bucket = lambda b, off, cnt: (order.append(b), inner(b, off, cnt))
                eng.set_bucket_callback(ar._on_bucket)
            norms = []
            for step in range(3):
                eng.set_input(x)
                eng.forward(train=True)
                eng.backward(1e6, 1e-4)
                if ar is not None:
                    ar.step(eng, 1e-3)
                else:
                    eng.adamw_step(1e-3)
                norms.append(eng.last_grad_norm())
            torch.cuda.synchronize()
            if ar is not None:
                nb = eng.bucket_count()
                assert order[-nb:] == list(range(nb - 2)) + [nb - 1, nb - 2]
            outs.append((eng.state_dict(), norms))
            eng.close()
        (sa, na), (sb, nb_) = outs
        # float atomics (bias / <G,W> / power-iteration accumulation) make two runs of the SAME path differ in the
        # last bits, and Adam's first steps turn a sign flip of a near-zero gradient into a 2*lr difference on that
        # element: compare in the mean, not element by element
        np.testing.assert_allclose(na, nb_, rtol=1e-4)
        for k in sa:
            a, b = sa[k].astype(np.float64), sb[k].astype(np.float64)
            assert np.mean(np.abs(a - b)) <= 3e-4 * np.mean(np.abs(a)) + 1e-9, k
    finally:
        if created:
            dist.destroy_process_group()


def test_data_scaler_matches_reference_fixture(tmp_path, monkeypatch):
    """GPU data_scaler (per-node MinMaxScaler(-0.7, 0.7) fitted on the reference's seeded row sample,
    transform of the whole array) against the outputs recorded from the reference; then the HBM-resident variant
    feeding create_augmented_dataloaders without another conversion.  fp32 tolerance: 2e-6 absolute on values in
    [-0.7, 0.7] (the reference multiplies then adds in two roundings, the kernel uses one FMA)."""
    from modules import data_preprocess as dp
    monkeypatch.chdir(tmp_path)
    g = np.load(os.path.join(GOLD, "scaler.npz"))
    raw = g["raw"]
    P, T, N = raw.shape
    out, shape, sc = dp.data_scaler(raw.copy(), raw, T, N, 1)
    assert tuple(shape) == tuple(g["shape"]) and out.shape == raw.shape and out.dtype == np.float32
    np.testing.assert_array_equal(sc.data_min_, g["data_min"])
    np.testing.assert_array_equal(sc.data_max_, g["data_max"])
    np.testing.assert_allclose(sc.scale_, g["scale"], rtol=1e-6)
    np.testing.assert_allclose(sc.min_, g["offset"], rtol=1e-5, atol=1e-6)
    assert sc.scale_[3] == np.float32(1.4)                       # zero-range node: scale = range of the target / 1
    ref = g["scaled"]
    assert np.abs(out - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    assert ref.max() > 100                                      # the outlier outside the sampled range is NOT clipped
    import pickle
    sk = pickle.load(open("model_save/scaler.pkl", "rb"))        # our own file, written a few lines above
    np.testing.assert_allclose(sk.transform(raw[0]), out[0], atol=5e-6)
    # resident variant: engine layout [P][T][N] bf16, consumed by the loaders as is
    N2 = 64
    raw2 = np.tile(raw, (1, 1, N2 // N))[:8, :12]
    dev, _, sc2 = dp.data_scaler(raw2.copy(), raw2, 12, N2, 1, device_dataset=True, compute_dtype="bf16")
    host, _, _ = dp.data_scaler(raw2.copy(), raw2, 12, N2, 1)
    want = torch.from_numpy(host).to(torch.bfloat16)
    got = dev.buf.view(torch.bfloat16).view(8, 12, N2).cpu()
    assert torch.equal(got, want)
    assert dev.shape == (8, N2, 12) and torch.equal(dev[3].cpu(), want[3].T.float())
    tl, vl = aug.create_augmented_dataloaders(dev, 2, load_all=True)
    from simulgen_vae_amd.engine import Engine
    from tests.gpu_common import make_cfg as mk
    cfg = mk(dict(latent_dim=32, hierarchical_dim=8, enc=[32, 16, 8, 8], num_node=N2, num_time=12))
    eng = Engine(cfg, max_batch=2, compute_dtype="bf16")
    assert tl.resident(eng).data_ptr() == dev.buf.data_ptr()
    eng.close()


def test_native_rccl_path_equals_plain_step():
    """include/sgvae.h native RCCL path (sgv_rccl_*, sgv_set_rccl, sgv_allreduce_grads) on a one-rank communicator: the
    engine issues the bucket all-reduces itself inside backward, sgv_adamw_step / sgv_backward_step order the waits; the
    state after three steps must be that of the plain step.  Also: the stream-ordered whole-arena sgv_allreduce_grads
    leaves one-rank gradients unchanged, and the two data-parallel registrations exclude each other."""
    import torch.distributed as dist
    from modules.train import NativeAllReduce
    from simulgen_vae_amd.engine import Engine, SgvError
    from simulgen_vae_amd.init import init_state
    from tests.gpu_common import G1
    cfg = make_cfg(G1)
    B = 4
    x = torch.from_numpy(synthetic_samples(5, range(B), cfg.num_node, cfg.num_time)).cuda()
    state = init_state(cfg, 11, reference_init=True)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29519", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    try:
        outs = []
        for mode in ("plain", "native", "native_fused"):
            eng = Engine(cfg, max_batch=B, compute_dtype="f32")
            eng.load_state(state)
            eng.seed(99)
            ar = NativeAllReduce.create(eng)[0] if mode != "plain" else None
            if ar is not None:
                # a one-rank NativeAllReduce leaves the engine on its single-GPU schedule (nothing to exchange); register the
                # communicator by hand so that the engine's own bucket / event / wait plumbing is what runs here
                assert ar.single
                eng.set_rccl(ar.comm, ar.stream.cuda_stream)
            norms = []
            for step in range(3):
                eng.set_input(x)
                eng.forward(train=True)
                if mode == "native_fused":
                    eng.backward_step(1e6, 1e-4, 1e-3)
                else:
                    eng.backward(1e6, 1e-4)
                    if ar is not None:
                        ar.step(eng, 1e-3)
                    else:
                        eng.adamw_step(1e-3)
                norms.append(eng.last_grad_norm())
            torch.cuda.synchronize()
            if mode == "native":
                with pytest.raises(SgvError, match="communicator is registered"):
                    eng.set_bucket_callback(lambda b, off, cnt: None)
                g0 = eng.grad("decoder.decoder_residual_blocks.1.seq.3.weight_orig").copy()
                eng.allreduce_grads(ar.comm, ar.stream.cuda_stream)
                eng.allreduce_grads(ar.comm)                       # on the engine's own stream
                torch.cuda.synchronize()
                np.testing.assert_array_equal(eng.grad("decoder.decoder_residual_blocks.1.seq.3.weight_orig"), g0)
            outs.append((eng.state_dict(), norms))
            if ar is not None:
                eng.set_rccl(None, None)
                ar.close()
            eng.close()
        (sa, na) = outs[0]
        for sb, nb_ in outs[1:]:
            np.testing.assert_allclose(na, nb_, rtol=1e-4)
            for k in sa:
                a, b = sa[k].astype(np.float64), sb[k].astype(np.float64)
                assert np.mean(np.abs(a - b)) <= 3e-4 * np.mean(np.abs(a)) + 1e-9, k
    finally:
        if created:
            dist.destroy_process_group()


def _two_rank_worker(rank, world, port, out_dir, payload="f32", ahead=False):
    """One data-parallel rank of test_two_rank_data_parallel_steps: real engine on cuda:0, gloo group (RCCL refuses two
    ranks on one device), modules.train.GradAllReduce exactly as train() uses it."""
    import torch.distributed as dist
    from modules.train import GradAllReduce
    from simulgen_vae_amd.engine import Engine
    from simulgen_vae_amd.init import init_state
    from tests.gpu_common import G1
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = make_cfg(G1)
    B = 2
    x = torch.from_numpy(synthetic_samples(5, range(rank * B, (rank + 1) * B), cfg.num_node, cfg.num_time)).cuda()
    eng = Engine(cfg, max_batch=B, compute_dtype="f32")
    eng.load_state(init_state(cfg, 11, reference_init=True))
    eng.seed(100 + rank)
    ar = GradAllReduce(eng, payload=payload)
    assert (ar.flat_lp is not None) == (payload == "bf16") and ar.early
    norms = []
    for step in range(3):
        eng.set_input(x)
        eng.forward(train=True)
        if ahead:
            # what train() and bench.py call: weight buckets travel with their <G,W> scalars and are updated on the optimizer
            # stream from inside the bucket callbacks, under the rest of backward
            ar.backward_step(eng, 1e6, 1e-4, 1e-3)
        else:
            eng.backward(1e6, 1e-4)
            ar.step(eng, 1e-3)
        norms.append(eng.last_grad_norm())
    torch.cuda.synchronize()
    sd = eng.state_dict()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), norms=np.asarray(norms), **{k.replace(".", "__"): v for k, v in sd.items()})
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ahead", [False, True], ids=["step_after_backward", "updates_under_backward"])
@pytest.mark.parametrize("payload", ["f32", "bf16"])
def test_two_rank_data_parallel_steps(tmp_path, payload, ahead):
    """SURVEY 8(e) with two real ranks: two processes, each with its own engine and its own shard / noise seed, average
    their gradient buckets through modules.train.GradAllReduce (bucket callbacks during backward, bucket-ranged AdamW).
    Expected state: the same three steps in ONE process, where the two shards' gradient arenas are averaged by hand
    before AdamW.  Both ranks must end with the same parameters.
    payload "bf16": the weight buckets travel as bf16 copies (sgv_set_grad_payload); the hand average then rounds each
    rank's weight-bucket gradients to bf16, adds them in bf16 and halves (what the collective computes), the small bucket
    stays fp32 -- same tolerance, because the expectation models the wire format.
    ahead: the ranks run GradAllReduce.backward_step (every weight bucket averaged together with the <G,W> scalars of its conv
    layers and updated on the optimizer stream from inside its callback, sgv_adamw_bucket_async); same expectation, since the
    update of a bucket does not depend on when it runs."""
    import socket
    import torch.multiprocessing as mp
    from simulgen_vae_amd.engine import Engine
    from simulgen_vae_amd.init import init_state
    from modules.train import _DevArray
    from tests.gpu_common import G1
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), payload, ahead), nprocs=2, join=True)
    got = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(2)]

    cfg = make_cfg(G1)
    B = 2
    engs, flats, xs = [], [], []
    for r in range(2):
        e = Engine(cfg, max_batch=B, compute_dtype="f32")
        e.load_state(init_state(cfg, 11, reference_init=True))
        e.seed(100 + r)
        ptr, n = e.grad_buffer()
        engs.append(e)
        flats.append(torch.as_tensor(_DevArray(ptr, n), device="cuda"))
        xs.append(torch.from_numpy(synthetic_samples(5, range(r * B, (r + 1) * B), cfg.num_node, cfg.num_time)).cuda())
    norms = []
    ranges = []
    engs[0].set_bucket_callback(lambda b, off, cnt: ranges.append((b, off, cnt)))
    for step in range(3):
        for e, x in zip(engs, xs):
            e.set_input(x)
            e.forward(train=True)
            e.backward(1e6, 1e-4)
        if step == 0:
            engs[0].set_bucket_callback(None)
            assert sorted(b for b, _, _ in ranges) == list(range(engs[0].bucket_count()))
        mean = (flats[0] + flats[1]) * 0.5
        if payload == "bf16":
            small = engs[0].bucket_count() - 1
            for b, off, cnt in ranges:
                if b != small:
                    sl = slice(off, off + cnt)
                    mean[sl] = ((flats[0][sl].bfloat16() + flats[1][sl].bfloat16()) * 0.5).float()
        for e, f in zip(engs, flats):
            f.copy_(mean)
            e.adamw_step(1e-3)
        norms.append(engs[0].last_grad_norm())
    torch.cuda.synchronize()
    want = engs[0].state_dict()
    for e in engs:
        e.close()
    # BITWISE: the step has no floating-point atomics, the mean of two ranks is one addition and one exact halving on either
    # side, and nothing in the data-parallel schedule may change a value.  (Until round 2 this was a 3e-4 tolerance, which hid a
    # real hazard: <G,W> of the Linear heads was computed at the end of backward from gradients the fp32 collective was already
    # reducing in place -- replicas stayed identical, the value depended on timing.  tests/micro/ddp_flake_diag.py found it.)
    np.testing.assert_array_equal(got[0]["norms"], norms)
    np.testing.assert_array_equal(got[1]["norms"], norms)
    for k, w in want.items():
        a, b = got[0][k.replace(".", "__")], got[1][k.replace(".", "__")]
        assert np.array_equal(a, w) and np.array_equal(b, w), k


@pytest.mark.parametrize("path,payload", [("torch", "f32"), ("torch", "bf16"), ("native", "f32"), ("native", "bf16"), ("torch-wire", "bf16")])
def test_one_rank_forced_collectives_through_rccl(monkeypatch, path, payload):
    """One-GPU rehearsal of the N > 1 step with REAL RCCL calls (SGV_FORCE_COLLECTIVE=1: a one-rank group normally issues no
    collective at all, so nothing else on a one-GPU box runs ncclAllReduce(ncclAvg) on the fp32 arena / the bf16 wire copy,
    the pack / unpack kernels around it and the stream-side waits of the bucket-ranged AdamW).  AVG over one rank is the
    identity on what travels: the expected state is the plain step with, for payload bf16, every weight bucket rounded to
    bf16 before AdamW (the small bucket stays fp32).  Both issue paths: torch.distributed (bucket callback) and the
    engine's own communicator (sgv_set_rccl)."""
    import torch.distributed as dist
    from modules.train import GradAllReduce, NativeAllReduce, _DevArray
    from simulgen_vae_amd.engine import Engine
    from simulgen_vae_amd.init import init_state
    from tests.gpu_common import G1
    monkeypatch.setenv("SGV_FORCE_COLLECTIVE", "1")
    monkeypatch.setenv("SGV_GRAD_PAYLOAD", payload)
    if path == "torch-wire":      # released buckets gathered and packed on the engine's wire stream, collectives issued from there
        monkeypatch.setenv("SGV_DDP_WIRE", "1")
        path = "torch"
    cfg = make_cfg(G1)
    B = 4
    x = torch.from_numpy(synthetic_samples(5, range(B), cfg.num_node, cfg.num_time)).cuda()
    state = init_state(cfg, 11, reference_init=True)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29523", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    try:
        outs = []
        for mode in ("expect", path):
            eng = Engine(cfg, max_batch=B, compute_dtype="f32")
            eng.load_state(state)
            eng.seed(99)
            ar = None
            if mode == "torch":
                ar = GradAllReduce(eng)
                assert not ar.single and (ar.flat_lp is not None) == (payload == "bf16")
                assert (ar.wire is not None) == (os.environ.get("SGV_DDP_WIRE") == "1")
            elif mode == "native":
                ar = NativeAllReduce.create(eng)[0]
                assert not ar.single
            ranges = []
            if mode == "expect":
                ptr, n = eng.grad_buffer()
                flat = torch.as_tensor(_DevArray(ptr, n), device="cuda")
                eng.set_bucket_callback(lambda b, off, cnt: ranges.append((b, off, cnt)))
            norms = []
            for step in range(3):
                eng.set_input(x)
                eng.forward(train=True)
                if ar is not None:
                    ar.backward_step(eng, 1e6, 1e-4, 1e-3)
                else:
                    eng.backward(1e6, 1e-4)
                    if step == 0:
                        eng.set_bucket_callback(None)
                    if payload == "bf16":
                        small = eng.bucket_count() - 1
                        for b, off, cnt in ranges:
                            if b != small:
                                flat[off:off + cnt] = flat[off:off + cnt].bfloat16().float()
                    eng.adamw_step(1e-3)
                norms.append(eng.last_grad_norm())
            torch.cuda.synchronize()
            outs.append((eng.state_dict(), norms))
            if mode == "native":
                eng.set_rccl(None, None)
                ar.close()
            eng.close()
        (sa, na), (sb, nb_) = outs
        assert all(np.isfinite(nb_))
        np.testing.assert_allclose(na, nb_, rtol=1e-4)
        for k in sa:
            a, b = sa[k].astype(np.float64), sb[k].astype(np.float64)
            assert np.mean(np.abs(a - b)) <= 3e-4 * np.mean(np.abs(a)) + 1e-9, k
    finally:
        if created:
            dist.destroy_process_group()


def test_make_allreduce_picks_the_engine_issued_path_and_falls_back_together(monkeypatch):
    """modules.train.make_allreduce on an RCCL group: the engine-issued path (own communicator on the engine's probed
    communication stream) by default; if setting it up fails on a rank, the ranks agree (MIN all-reduce of a flag) to take the
    callback path; SGV_DDP_NATIVE=0 / 1 force either.  One-rank group: what can be checked on one card is the choice itself, that
    the chosen object runs a step, and that the communication stream is the engine's (sgv_comm_stream)."""
    import torch.distributed as dist
    import modules.train as T
    from simulgen_vae_amd.engine import Engine
    from simulgen_vae_amd.init import init_state
    from tests.gpu_common import G1
    monkeypatch.delenv("SGV_DDP_NATIVE", raising=False)
    cfg = make_cfg(G1)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29527", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        eng = Engine(cfg, max_batch=2, compute_dtype="f32")
        eng.load_state(init_state(cfg, 11, reference_init=True))
        x = torch.from_numpy(synthetic_samples(5, range(2), cfg.num_node, cfg.num_time)).cuda()
        ar = T.make_allreduce(eng)
        assert isinstance(ar, T.NativeAllReduce) and ar.stream.cuda_stream == eng.comm_stream()
        info = ar.info()
        assert info["ddp_path"] == "native" and info["rccl_nranks"] == 1 and info["torch_world"] == 1 and info["buckets"] == eng.bucket_count()
        eng.set_input(x); eng.forward(train=True); ar.backward_step(eng, 1e6, 1e-4, 1e-3)
        assert np.isfinite(eng.last_grad_norm())
        ar.close()
        monkeypatch.setenv("SGV_DDP_NATIVE", "0")
        assert isinstance(T.make_allreduce(eng), T.GradAllReduce)
        monkeypatch.delenv("SGV_DDP_NATIVE")

        # the set-up failing in a way the ranks agree on (NativeAllReduce.create returns (None, reason) everywhere)
        monkeypatch.setattr(T.NativeAllReduce, "create", classmethod(lambda cls, e, g=None: (None, "no RCCL here")))
        ar = T.make_allreduce(eng)
        assert isinstance(ar, T.GradAllReduce)
        assert ar.info()["ddp_path"] == "torch"
        monkeypatch.setenv("SGV_DDP_NATIVE", "1")
        with pytest.raises(RuntimeError, match="no RCCL here"):
            T.make_allreduce(eng)
        monkeypatch.delenv("SGV_DDP_NATIVE")
        eng.set_input(x); eng.forward(train=True); ar.backward_step(eng, 1e6, 1e-4, 1e-3)
        assert np.isfinite(eng.last_grad_norm())
        eng.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("chunks", [2, 1])
@pytest.mark.parametrize("early", [1, 0])
@pytest.mark.parametrize("payload", ["f32", "bf16"])
def test_engine_issued_collectives_cover_every_gradient_exactly_once(payload, early, chunks, monkeypatch):
    """N > 1 correctness of the engine-issued data-parallel step without a second GPU.  A one-rank ncclAvg is the identity, so a
    range that is never reduced -- or reduced twice -- cannot show in any one-rank run.  Here ncclAllReduce is replaced by a test
    double that multiplies the range it is handed by k in place (sgv_test_fake_collective; fp32 and bf16 ranges): with k = 2 a step
    at (alpha, beta) must leave BITWISE the state of a step with k = 1 at (2 alpha, 2 beta) -- backward is linear in (alpha, beta)
    and a factor of two is exact in fp32 and bf16 -- iff every gradient element, every <G,W> slot and the small zone went through
    exactly one collective before the update that reads it.  All modes of the path: fp32 / bf16 wire format, per-bucket updates
    under backward on / off, the first layer's bucket exchanged in chunks of its weight-gradient GEMM / in one piece (preset
    widths, N = 4096: the chunked exchange needs 1024 output channels)."""
    import ctypes as C
    from simulgen_vae_amd.engine import Engine, load_library
    from simulgen_vae_amd.init import init_state
    lib = load_library()
    G2 = dict(latent_dim=32, hierarchical_dim=8, enc=[1024, 512, 256, 128], num_node=4096, num_time=32)     # preset widths (fixture g2's shape)
    monkeypatch.setenv("SGV_GRAD_PAYLOAD", payload)
    monkeypatch.setenv("SGV_DDP_LAST_CHUNKS", str(chunks))
    monkeypatch.setenv("SGV_DDP_CHUNK_MIN_GF", "0")
    cfg = make_cfg(G2)
    B = 4
    x = torch.from_numpy(synthetic_samples(5, range(B), cfg.num_node, cfg.num_time)).cuda()
    state = init_state(cfg, 11, reference_init=True)
    names = ["encoder.encoder_blocks.0.module_list.0._seq.0.weight_orig", "encoder.encoder_blocks.0.module_list.0._seq.0.bias",
             "decoder.recon.0.weight_orig", "decoder.recon.1.weight", "decoder.decoder_residual_blocks.2.seq.3.weight_orig",
             "decoder.decoder_residual_blocks.0.seq.0.weight_orig", "encoder.xs_linear.1.weight_orig", "encoder.encoder_blocks.2.module_list.0._seq.1.bias",
             "decoder.condition_xz.0.2.weight_orig", "decoder.sequence_start.0.0.weight_orig"]
    outs, counts = [], []
    try:
        for k, mult in ((2.0, 1.0), (1.0, 2.0)):
            assert lib.sgv_test_fake_collective(k, None, None) == 0
            eng = Engine(cfg, max_batch=B, compute_dtype="bf16")
            eng.load_state(state)
            eng.seed(99)
            eng.set_option("ddp_early_adamw", early)
            if payload == "bf16":
                eng.set_grad_payload("bf16")
            eng.set_rccl(0x5eed, eng.comm_stream())         # any non-null handle: the double never looks at it
            norms = []
            for step in range(2):
                eng.set_input(x)
                eng.forward(train=True)
                eng.backward_step(1e6 * mult, 1e-4 * mult, 1e-3)
                norms.append(eng.last_grad_norm())
            torch.cuda.synchronize()
            calls, elems = C.c_long(), C.c_long()
            assert lib.sgv_test_fake_collective(k, C.byref(calls), C.byref(elems)) == 0
            counts.append((calls.value, elems.value))
            ptr, n = eng.grad_buffer()
            sd = eng.state_dict()
            outs.append((norms, {kk: sd[kk] for kk in sd if kk in names or kk.endswith("weight_u")}, n))
            eng.set_rccl(None, None)
            eng.close()
    finally:
        lib.sgv_test_fake_collective(0.0, None, None)
    (na, sa, n_arena), (nb, sb, _) = outs
    assert counts[0] == counts[1] and counts[0][0] > 0
    # every arena element went into a collective exactly once per step (the arena is padded per bucket: >=)
    assert counts[0][1] >= 2 * n_arena - 64 * 2 and counts[0][1] <= 2 * n_arena, (counts, n_arena)
    assert na == nb, (na, nb)                       # gradient norm of (2 alpha, 2 beta) == norm of the doubled gradients, bitwise
    assert set(names) <= set(sa)
    for kk in sa:
        assert np.array_equal(sa[kk], sb[kk]), kk


def test_fused_backward_step_equals_separate_calls():
    """sgv_backward_step (AdamW of finished buckets started on the side stream under the rest of backward) leaves the
    state of sgv_backward + sgv_adamw_step; gradients stay exportable afterwards.  fp32 compute, compared in the mean
    (float-atomic ordering differs between any two runs)."""
    from simulgen_vae_amd.engine import Engine
    from simulgen_vae_amd.init import init_state
    from tests.gpu_common import G2
    cfg = make_cfg(G2)
    B = 2
    x = torch.from_numpy(synthetic_samples(5, range(B), cfg.num_node, cfg.num_time)).cuda()
    state = init_state(cfg, 11, reference_init=True)
    outs = []
    for mode in ("separate", "fused"):
        eng = Engine(cfg, max_batch=B, compute_dtype="f32")
        eng.load_state(state)
        eng.seed(99)
        norms = []
        for step in range(3):
            eng.set_input(x)
            eng.forward(train=True)
            if mode == "fused":
                eng.backward_step(1e6, 1e-4, 1e-3)
            else:
                eng.backward(1e6, 1e-4)
                eng.adamw_step(1e-3)
            norms.append(eng.last_grad_norm())
        g = eng.grad("decoder.decoder_residual_blocks.2.seq.3.weight_orig")
        torch.cuda.synchronize()
        outs.append((eng.state_dict(), norms, g))
        eng.close()
    (sa, na, ga), (sb, nb_, gb) = outs
    np.testing.assert_allclose(na, nb_, rtol=1e-4)
    assert relerr(gb, ga) < 1e-3
    for k in sa:
        a, b = sa[k].astype(np.float64), sb[k].astype(np.float64)
        # Adam's normalisation turns float-atomic ordering noise on near-zero gradients into +-lr steps: compare in the mean
        assert np.mean(np.abs(a - b)) <= 3e-4 * np.mean(np.abs(a)) + 1e-9, k


def test_end_to_end_pipeline_of_the_reference_main(tmp_path, monkeypatch):
    """The stages SimulGen-VAE.py chains (SimulGen-VAE.py:267-473), through the mirrors only: raw [P,T,N] array ->
    data_scaler (HBM-resident) -> create_augmented_dataloaders -> train (VAE) -> export_latents (model_save/*.npy) ->
    LatentConditionerImg + train_latent_conditioner on images with the exported latents as targets.  Checks that every
    hand-off has the reference's shapes / files; values are covered by the per-stage parity tests."""
    from modules import data_preprocess as dp
    from modules import utils as U
    from modules.latent_conditioner_model_cnn import LatentConditionerImg
    from modules.latent_conditioner import train_latent_conditioner
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    random.seed(0)
    np.random.seed(0)
    enc = [64, 32, 16, 8]
    P, T, N, B, latent_end, latent = 8, 16, 512, 4, 32, 8
    raw = np.random.default_rng(1).standard_normal((P, T, N)).astype(np.float32) * 3.0
    dev, shape, scaler = dp.data_scaler(raw.copy(), raw, T, N, 1, device_dataset=True, compute_dtype="bf16")
    assert dev.shape == (P, N, T) and tuple(shape) == (T, N)
    tl, vl = aug.create_augmented_dataloaders(dev, B, load_all=True)
    out = tr.train(4, B, tl, vl, 1e-3, enc, enc[::-1], N, latent_end, latent, T, 1e6, "MSE", True, True)
    assert len(out) == 4 and all(len(a) == 4 for a in out) and np.all(np.isfinite(out[0]))
    m = torch.load("model_save/SimulGen-VAE", weights_only=False)          # our own pickle, written by train() above
    m.eval()
    x_all = np.stack([dev[i].cpu().numpy() for i in range(P)])             # [P, N, T] as the reference holds new_x_train
    lat, hier, rl = U.export_latents(m, x_all, enc, latent, latent_end)
    assert lat.shape == (P, latent_end) and hier.shape == (P, len(enc) - 1, latent) and np.all(np.isfinite(rl))
    assert os.path.exists("model_save/latent_vectors.npy") and os.path.exists("model_save/xs.npy") and os.path.exists("SimulGen-VAE_L2_loss.txt")
    # latent conditioner on synthetic 16x16 "images" with the exported latents as regression targets
    imgs = np.random.default_rng(2).random((P, 16 * 16)).astype(np.float32)
    y1 = np.load("model_save/latent_vectors.npy").astype(np.float32)
    y2 = np.load("model_save/xs.npy").astype(np.float32)
    batches = [(imgs[i:i + 4], y1[i:i + 4], y2[i:i + 4]) for i in range(0, P, 4)]
    lcm = LatentConditionerImg([16, 32, 32, 64, 64, 128], latent_end, (1, 16, 16), latent, len(enc) - 1, (16, 16), dropout_rate=0.2)
    val = train_latent_conditioner(3, batches, batches[:1], lcm, 1e-3, weight_decay=1e-5, is_image_data=True)
    assert np.isfinite(val) and os.path.exists("checkpoints/latent_conditioner.pth") and os.path.exists("model_save/LatentConditioner")
    lcm.eval()
    p1, p2 = lcm(imgs[:2])
    assert tuple(p1.shape) == (2, latent_end) and tuple(p2.shape) == (2, len(enc) - 1, latent)
    # the conditioner's predictions feed the VAE decoder (reference: latent_conditioner_e2e.py:371, utils.py:499)
    xh, _ = m.decoder(p1, [p2[:, i].contiguous() for i in range(p2.shape[1])], mode="fix")
    assert tuple(xh.shape) == (2, N, T) and bool(torch.isfinite(xh).all())
