"""CPU tests of the host-side mirror (simulgen-vae_amd/modules): config parsing, schedules, the
augmentation draw order, and the data-parallel definition with 2 gloo ranks."""
import os
import random
import socket

import numpy as np
import pytest
import torch

import simulgen_vae_amd
from simulgen_vae_amd.init import init_state, synthetic_eps, synthetic_samples
from simulgen_vae_amd.spec import VAEConfig, param_spec

simulgen_vae_amd.install_reference_api()
from modules import augmentation as aug  # noqa: E402
from modules import train as tr  # noqa: E402
from modules import utils as ut  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")

CONDITION = """Common params
Dim1		8 # number of parameters
Dim2		32 # number of timesteps
Dim3		4096 # num nodes
num_var 1
RESERVED	0
RESERVED	7
'
%LSH-VAE parameters
Training_epochs	4
Batch_size	4
LearningR	0.001
Latent_dim	8	# Hierarchical latent Dim1
Latent_dim_end	32 # Main latent Dim1
Loss_type	3	# 1: MSE, 2, MAE, 3: smoothL1, 4: Huber
Stretch	0
alpha		1000000
Recon_iter	1
Dim2_red		32
Dim3_start      0
Dim3_end		4096
'
%LatentConditioner
num_param	0
param_dir	/images
n_epoch	5000
latent_conditioner_lr	0.001	# comment
latent_conditioner_batch	64
input_type	image	#image, csvs
param_data_type .png
"""


def test_condition_file_and_preset(tmp_path):
    f = tmp_path / "condition.txt"
    f.write_text(CONDITION)
    p = ut.parse_condition_file(str(f))
    assert p["RESERVED"] == "7"                  # last duplicate key wins
    assert "%LSH-VAE" not in p and "Common" in p  # section markers skipped; 'Common params' is a key/value line
    c = ut.parse_training_parameters(p)
    assert (c["num_param"], c["num_time"], c["num_node"], c["batch_size"], c["n_epochs"]) == (8, 32, 4096, 4, 4)
    assert c["latent_dim"] == 8 and c["latent_dim_end"] == 32 and c["alpha"] == 1000000 and c["LR"] == 1e-3
    assert ut.LOSS_NAMES[c["loss_type"]] == "smoothL1"
    assert c["latent_conditioner_weight_decay"] == 1e-4 and c["use_e2e_training"] == 0   # defaults
    del p["alpha"]
    with pytest.raises(KeyError):
        ut.parse_training_parameters(p)
    pre = tmp_path / "preset.txt"
    pre.write_text("data_No, init_beta_divisior, num_filter_enc, latent_conditioner_filter\n1\n0\n1024 512 256 128\n32 64 128\n")
    r = ut.read_preset(str(pre))
    assert r["num_filter_enc"] == [1024, 512, 256, 128] and r["data_No"] == 1


def test_schedules_match_reference_golden():
    g = np.load(os.path.join(GOLD, "schedules.npz"))
    for E in (4, 8, 20, 40):
        np.testing.assert_allclose([tr.cosine_warm_restarts_lr(1e-3, E, e) for e in range(E)], g[f"lr_E{E}"], rtol=1e-9)
    for E in (4, 10, 20):
        w = tr.WarmupKLLoss(E, 1e-4, int(E * 0.3), int(E * 0.8), 1)
        np.testing.assert_allclose([w.get_loss(e, [0.0])[0] for e in range(E)], g[f"beta_E{E}"], rtol=1e-12)
    with pytest.raises(ValueError):
        tr.cosine_warm_restarts_lr(1e-3, 3, 0)


def test_augmentation_draw_order_matches_reference(monkeypatch):
    """Replay the reference's recorded random draws through AugmentedDataset.plan: same decisions,
    scale factors, mixup partners and lambdas, and (with the recorded noise) the same samples."""
    g = np.load(os.path.join(GOLD, "augment.npz"))
    P, N, T = g["shape"]
    data = synthetic_samples(20251003, range(P), N, T)
    rand, randint, beta, noise = list(g["rand"]), list(g["randint"]), list(g["beta"]), list(g["noise"])

    class R:
        random = staticmethod(lambda: rand.pop(0))
        randint = staticmethod(lambda a, b: int(randint.pop(0)))
        getrandbits = staticmethod(lambda k: 12345)

    monkeypatch.setattr(aug, "random", R)
    monkeypatch.setattr(aug.np.random, "beta", lambda a, b: beta.pop(0))
    ds = aug.AugmentedDataset(data, load_all=False)
    for i in range(P):
        # the reference draws the noise decision first; getrandbits is ours and consumes nothing recorded
        seed, scale, mix, lam = ds.plan(i)
        x = data[i].copy()
        if seed:
            x = x + noise.pop(0) * np.float32(0.05)
        x = x * np.float32(scale)
        if mix >= 0:
            x = lam * x + (1 - lam) * data[mix]
        np.testing.assert_allclose(x, g["out"][i], rtol=1e-6, atol=1e-7)
    assert not rand and not randint and not beta and not noise


def test_loader_split_and_sharding():
    torch.manual_seed(0)
    random.seed(0)
    x = np.zeros((10, 8, 4), np.float32)
    tl, vl = aug.create_augmented_dataloaders(x, batch_size=3, load_all=False)
    assert len(tl.indices) == 8 and len(vl.indices) == 2 and not set(tl.indices) & set(vl.indices)
    plans = list(tl.batch_plans())
    assert [len(p[0]) for p in plans] == [3, 3, 2]
    assert sorted(i for p in plans for i in p[0]) == sorted(tl.indices)
    assert all(p[1] == [0] * len(p[0]) and p[3] == [-1] * len(p[0]) for p in vl.batch_plans())   # no aug on val
    idx = list(range(11))
    a, b = ut.shard_indices(idx, 0, 2, 3), ut.shard_indices(idx, 1, 2, 3)
    assert len(a) == len(b) == 5 and not set(a) & set(b)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    from oracle import vae_oracle as vo
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    enc = [32, 16, 8, 8]
    cfg = VAEConfig(32, 8, enc, enc[::-1], 72, 10, "MSE", True)
    B = 4
    x = synthetic_samples(20251003, range(B), cfg.num_node, cfg.num_time)
    eps = synthetic_eps(1234, 0, cfg, B)
    sl = slice(rank * B // world, (rank + 1) * B // world)
    m = vo.OracleVAE(cfg, init_state(cfg, 7))
    m.forward(x[sl], [e[sl] for e in eps])
    grads = m.backward(1e6, 1e-4)
    names = [e.name for e in param_spec(cfg) if grads.get(e.name) is not None]
    flat = torch.from_numpy(np.concatenate([grads[n].ravel() for n in names]))
    # bucketed mean all-reduce exactly as modules.train.GradAllReduce issues it (offset/count slices)
    nb = 5
    edges = np.linspace(0, flat.numel(), nb + 1).astype(int)
    works = [dist.all_reduce(flat[edges[i]:edges[i + 1]], op=dist.ReduceOp.SUM, async_op=True) for i in range(nb)]
    for w in works:
        w.wait()
    flat /= world
    if rank == 0:
        np.save(out, flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_mean_gradient_equals_full_batch(tmp_path):
    """DDP definition: mean over ranks of per-shard gradients == single-process gradient of
    the concatenated batch (every loss term is a batch mean, GroupNorm is per-sample)."""
    import torch.multiprocessing as mp
    from oracle import vae_oracle as vo
    out = str(tmp_path / "g.npy")
    mp.spawn(_ddp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    enc = [32, 16, 8, 8]
    cfg = VAEConfig(32, 8, enc, enc[::-1], 72, 10, "MSE", True)
    x = synthetic_samples(20251003, range(4), cfg.num_node, cfg.num_time)
    eps = synthetic_eps(1234, 0, cfg, 4)
    m = vo.OracleVAE(cfg, init_state(cfg, 7))
    m.forward(x, eps)
    grads = m.backward(1e6, 1e-4)
    ref = np.concatenate([grads[e.name].ravel() for e in param_spec(cfg) if grads.get(e.name) is not None])
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-5


def _loader_worker(rank, world, port, out_dir):
    """Unseeded ranks (each process keeps its own urandom-seeded `random` / torch generators, as under torchrun)."""
    import os
    import random
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    random.seed(1000 + 17 * rank)          # deliberately DIFFERENT global streams per rank
    torch.manual_seed(2000 + 31 * rank)
    from simulgen_vae_amd.modules.augmentation import create_augmented_dataloaders
    x = np.zeros((40, 6, 4), np.float32)
    tl, vl = create_augmented_dataloaders(x, batch_size=4, load_all=False)
    epochs = []
    for _ in range(3):
        epochs.append([b for b, *_ in tl.batch_plans()])       # also consumes augmentation draws (rank-dependent amounts)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([tl.indices, vl.indices, epochs], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_loaders_share_split_and_disjoint_shards(tmp_path):
    """without any external seeding, both ranks must hold the SAME train/validation split and, every epoch,
    disjoint shards of the SAME permutation (rank 0's split and shuffle seed are broadcast; the permutation does not come
    from the process-global `random`, whose state diverges between ranks)."""
    import torch.multiprocessing as mp
    mp.spawn(_loader_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npy", allow_pickle=True)
    r1 = np.load(tmp_path / "r1.npy", allow_pickle=True)
    assert list(r0[0]) == list(r1[0]) and list(r0[1]) == list(r1[1])             # identical split
    assert not set(r0[0]) & set(r0[1]) and len(r0[0]) == 32 and len(r0[1]) == 8
    seen = []
    for e0, e1 in zip(r0[2], r1[2]):
        a = [i for b in e0 for i in b]
        b = [i for bb in e1 for i in bb]
        assert len(a) == len(b) == 16 and not set(a) & set(b)                    # disjoint, equal-sized shards
        assert set(a) | set(b) == set(r0[0])                                      # together: the whole training set
        seen.append(tuple(a))
    assert len(set(seen)) == 3                                                    # a new permutation every epoch


def test_input_pipeline_host_side_matches_reference_fixture():
    """SURVEY 8(f) N3, host half: reduce_dataset and the seeded row sampling of data_scaler against the fixture
    recorded from the reference (tests/golden/gen_fixtures.py scaler)."""
    import os
    from simulgen_vae_amd.modules import data_preprocess as dp
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "scaler.npz"))
    raw = g["raw"]
    P, T, N = raw.shape
    nt, red, nn = dp.reduce_dataset(raw.copy(), 150, 8, P, T, 4, 12)
    assert (nt, nn) == tuple(g["reduced_meta"]) and red.dtype == np.float64
    np.testing.assert_array_equal(red.astype(np.float32), g["reduced"])
    same = dp.reduce_dataset(raw, T, N, P, T, 0, N)
    assert same[1] is raw
    idx, max_samples, stride = dp._sample_rows(P * T)
    assert max_samples == 1200 and stride == 10 and len(set(idx.tolist())) == 1200
    rows = raw.reshape(-1, N)[idx]
    np.testing.assert_allclose(rows.min(0), g["data_min"], rtol=0, atol=0)     # the reference saw exactly these rows
    np.testing.assert_allclose(rows.max(0), g["data_max"], rtol=0, atol=0)


def test_latent_conditioner_lr_schedule_matches_torch_schedulers():
    """lc_learning_rate (closed form) vs the reference's LinearLR + CosineAnnealingLR pair stepped exactly as
    latent_conditioner.py:196-209,358-361 does (warm-up scheduler for epoch < 100, cosine afterwards)."""
    import torch
    from simulgen_vae_amd.modules.latent_conditioner import lc_learning_rate
    for epochs, base in ((300, 1e-3), (101, 5e-4), (150, 2e-3)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=base, weight_decay=1e-4)
        warm = torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, total_iters=100)
        main = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=epochs - 100, eta_min=1e-8)
        for epoch in range(epochs):
            assert abs(opt.param_groups[0]["lr"] - lc_learning_rate(base, epochs, epoch)) <= 1e-9 * base + 1e-15, (epochs, epoch)
            opt.step()
            (warm if epoch < 100 else main).step()


def test_e2e_cosine_schedule_and_latent_scaler(tmp_path):
    """SURVEY 8(f) N4 host side: cosine_lr (closed form) vs CosineAnnealingLR(T_max = epochs, eta_min = 1e-8) stepped once
    per epoch (latent_conditioner_e2e.py:141-146,516); latent_conditioner_scaler vs sklearn's MinMaxScaler on 2-D and 3-D
    arrays, the pickle it leaves behind, and its empty-input error (data_preprocess.py:167-195)."""
    import pickle
    import torch
    from sklearn.preprocessing import MinMaxScaler
    from simulgen_vae_amd.modules.latent_conditioner_e2e import cosine_lr, load_scaler
    from simulgen_vae_amd.modules.data_preprocess import latent_conditioner_scaler
    for epochs, base in ((7, 1e-3), (120, 3e-4)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=base)
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=epochs, eta_min=1e-8)
        for epoch in range(epochs + 1):
            assert abs(opt.param_groups[0]["lr"] - cosine_lr(base, epochs, epoch)) <= 1e-9 * base + 1e-15, (epochs, epoch)
            opt.step()
            sch.step()
    rng = np.random.default_rng(4)
    a2, a3 = rng.standard_normal((11, 5)) * 3, rng.standard_normal((11, 3, 4))
    s2, sc2 = latent_conditioner_scaler(a2, str(tmp_path / "a.pkl"))
    s3, sc3 = latent_conditioner_scaler(a3, str(tmp_path / "b.pkl"))
    ref = MinMaxScaler(feature_range=(-0.7, 0.7)).fit(a2)
    np.testing.assert_allclose(s2, ref.transform(a2), rtol=1e-12, atol=1e-12)
    assert s3.shape == a3.shape and abs(s3.min() + 0.7) < 1e-12 and abs(s3.max() - 0.7) < 1e-12
    with open(tmp_path / "b.pkl", "rb") as f:
        np.testing.assert_allclose(pickle.load(f).scale_, sc3.scale_)
    assert load_scaler(str(tmp_path / "missing.pkl")) is None
    with pytest.raises(ValueError, match="Empty data array"):
        latent_conditioner_scaler(np.zeros((0, 4)), str(tmp_path / "c.pkl"))


def test_use_checkpointing_true_is_accepted_and_forced_off_like_the_reference():
    """The reference accepts use_checkpointing and forces it to False (VAE_network.py:60,68); the mirror does the same and says
    so (no recompute path exists: every activation stays resident)."""
    from simulgen_vae_amd.modules.VAE_network import VAE
    enc = [32, 16, 8, 8]
    assert VAE(32, 8, enc, enc[::-1], 72, 10, use_checkpointing=False).use_checkpointing is False
    with pytest.warns(UserWarning, match="accepted and ignored"):
        m = VAE(32, 8, enc, enc[::-1], 72, 10, use_checkpointing=True)
    assert m.use_checkpointing is False
    # nn.Module.parameters(): the trainable tensors (no spectral-norm u / v buffers), e.g. for a parameter count
    n = sum(p.numel() for p in m.parameters())
    sd = m.state_dict()
    assert n == sum(v.numel() for k, v in sd.items() if not k.endswith(("weight_u", "weight_v"))) and n > 0


def test_decoder_signature_corners_are_refused_not_ignored():
    """decoder(z, xs=None) fails in the reference (broadcast of [B, latent] onto [B, C, T], decoder.py:179) and raises here;
    freeze_level >= 1 with mode='fix' (a cross-call latent cache, decoder.py:202-207) raises NotImplementedError instead of being
    dropped; neither needs a GPU to be refused."""
    import torch
    from simulgen_vae_amd.modules.VAE_network import VAE
    enc = [32, 16, 8, 8]
    m = VAE(32, 8, enc, enc[::-1], 72, 10)
    z = torch.zeros(2, 32)
    with pytest.raises(RuntimeError, match="xs=None"):
        m.decoder(z, None)
    with pytest.raises(NotImplementedError, match="freeze_level"):
        m.decoder(z, [torch.zeros(2, 8)] * 3, mode="fix", freeze_level=1)


def test_keyed_augmentation_plans_do_not_depend_on_the_world_size():
    """augmentation keyed by the position in the global shuffled list.  Two ranks' loaders (r::2 shards) must plan,
    for every global position, exactly what a one-rank loader with the same job seed plans -- sample index, noise seed, scale,
    mixup partner and lambda -- whatever the process-global random streams hold."""
    import random
    from simulgen_vae_amd.modules.augmentation import AugmentedDataset, ResidentLoader
    x = np.zeros((40, 6, 4), np.float32)
    idx = list(range(37))
    def plans(rank, world, bs, seed_global):
        random.seed(seed_global); np.random.seed(seed_global)
        ld = ResidentLoader(AugmentedDataset(x, False), idx, bs, shuffle=True, augment=True, rank=rank, world=world, shuffle_seed=4242)
        out = []
        for _ in range(2):
            out.append([list(zip(*item)) for item in ld.batch_plans()])
        return out
    one = plans(0, 1, 8, 1)
    r0, r1 = plans(0, 2, 4, 77), plans(1, 2, 4, 999)
    for e in range(2):
        flat1 = [p for batch in one[e] for p in batch]
        f0 = [p for batch in r0[e] for p in batch]
        f1 = [p for batch in r1[e] for p in batch]
        assert len(f0) == len(f1) == 18               # 37 // 2 per rank (drop-last to equal shards)
        for k in range(18):
            assert f0[k] == flat1[2 * k] and f1[k] == flat1[2 * k + 1], (e, k)
        assert any(p[1] for p in f0) and any(p[3] >= 0 for p in f0)      # noise seeds and mixup partners do occur


def test_lookahead_pairs_every_batch_with_its_successor():
    """modules/train.py's prefetch loop: (current, next) over the epoch's batch plans, next = None for the last batch; an empty
    epoch yields nothing (the caller then raises the reference's ZeroDivisionError)."""
    from simulgen_vae_amd.modules.train import _lookahead
    def pairs(seq):
        it = iter(seq)
        return list(_lookahead(it, next(it, None)))
    assert pairs([]) == []
    assert pairs(["a"]) == [("a", None)]
    assert pairs(["a", "b", "c"]) == [("a", "b"), ("b", "c"), ("c", None)]
