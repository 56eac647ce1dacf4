"""Engine parity on the GPU, through the C ABI: libsgvae.so vs (a) the golden fixtures the
reference itself produced and (b) the numpy oracle, on the same seeded inputs.

Tolerances: fp32 compute -> 5e-5 relative (max-norm) on activations/gradients, 1e-5 on scalars;
bf16 compute -> stated per check below (bf16 has 8 mantissa bits; accumulation is fp32)."""
import os

import numpy as np
import pytest

from tests.gpu_common import G0, G1, G2, GOLD, engine_step, make_cfg, oracle_step, rel_l2, relerr
from simulgen_vae_amd import engine as E
from simulgen_vae_amd.init import init_state, synthetic_eps, synthetic_samples
from simulgen_vae_amd.spec import param_spec
from oracle import vae_oracle as vo

pytestmark = pytest.mark.gpu


def _golden_run(tag, cfgd, small, lossfun, dtype, steps=3):
    import torch
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    alpha, beta, lr, sseed, dseed, eseed, B = g["meta"]
    B = int(B)
    cfg = make_cfg(cfgd, small, lossfun)
    eng = E.Engine(cfg, max_batch=B, compute_dtype=dtype)
    eng.load_state(init_state(cfg, int(sseed)))
    out = []
    for step in range(steps):
        x = synthetic_samples(int(dseed), range(step * B, (step + 1) * B), cfg.num_node, cfg.num_time)
        eps = synthetic_eps(int(eseed), step, cfg, B)
        sc, acts = engine_step(eng, cfg, x, eps, alpha, beta, want_acts=(step == 0))
        gn = eng.grad_norm()
        loss = alpha * sc["recon"] + beta * sum(sc["kls"])
        rec = dict(scalars=np.array([sc["recon"]] + sc["kls"] + [sc["mse"], loss, gn]), acts=acts)
        if step == 0:
            rec["grads"] = {e.name: eng.grad(e.name) for e in param_spec(cfg)
                            if e.kind in ("bias", "weight_orig", "gn_weight", "gn_bias")}
            rec["uv"] = {k: v for k, v in eng.state_dict().items() if k.endswith("_u") or k.endswith("_v")}
        eng.adamw_step(lr)
        out.append(rec)
    return cfg, eng, g, out


@pytest.mark.parametrize("tag,small", [("g0_small_MSE", True), ("g0_large_MSE", False)])
def test_fp32_matches_reference_golden(tag, small):
    cfg, eng, g, out = _golden_run(tag, G0, small, "MSE", "f32")
    r0 = out[0]
    np.testing.assert_allclose(r0["scalars"], g["scalars0"], rtol=2e-5)
    assert relerr(r0["acts"]["x_hat"], g["x_hat"]) < 1e-5
    nograd = set(g["nograd"].tolist())
    for k in g.files:
        if k.startswith("act."):
            assert relerr(r0["acts"][k[4:]], g[k]) < 1e-5, k
        elif k.startswith("uv1."):
            assert relerr(r0["uv"][k[4:]], g[k]) < 1e-5, k
        elif k.startswith("grad."):
            assert relerr(r0["grads"][k[5:]], g[k]) < 5e-5, k
    assert nograd == {k for k, v in r0["grads"].items() if v is None}
    np.testing.assert_allclose(out[1]["scalars"], g["scalars1"], rtol=5e-5)
    np.testing.assert_allclose(out[2]["scalars"], g["scalars2"], rtol=2e-4)
    sd = eng.state_dict()
    for e in param_spec(cfg):
        assert relerr(sd[e.name], g["s3." + e.name]) < 3e-4, e.name
    # eval-mode forward + mode='fix' decoder on the step-3 state
    import torch
    B = int(g["meta"][6])
    x = synthetic_samples(int(g["meta"][4]), range(100, 100 + B), cfg.num_node, cfg.num_time)
    eps = synthetic_eps(int(g["meta"][5]), 100, cfg, B)
    eng.set_input(torch.from_numpy(x).cuda())
    eng.set_eps([torch.from_numpy(e).cuda() for e in eps])
    sc = eng.forward(train=False)
    np.testing.assert_allclose([sc["recon"]] + sc["kls"] + [sc["mse"]], g["eval.scalars"], rtol=5e-4)
    xh = eng.xhat().cpu().numpy()
    assert relerr(xh, g["eval.x_hat"]) < 3e-4
    mu, lv, xs = eng.encode()
    assert relerr(mu, g["eval.mu"]) < 3e-4 and relerr(lv, g["eval.log_var"]) < 3e-4
    for i, v in enumerate(xs):
        assert relerr(v, g[f"eval.xs{i}"]) < 3e-4
    eng.set_eps([torch.from_numpy(e).cuda() for e in eps])
    eng.forward(train=False, fix=True)
    assert relerr(eng.xhat().cpu().numpy(), g["fix.x_hat"]) < 3e-4
    assert relerr(eng.activation("z", (B, cfg.latent_dim)), g["fix.z"]) < 3e-4
    eng.close()


@pytest.mark.parametrize("tag,cfgd,lossfun", [("g0_small_MAE", G0, "MAE"), ("g0_small_smoothL1", G0, "smoothL1"),
                                              ("g0_small_Huber", G0, "Huber"), ("g1_small_MSE", G1, "MSE")])
def test_fp32_matches_reference_norms(tag, cfgd, lossfun):
    cfg, eng, g, out = _golden_run(tag, cfgd, True, lossfun, "f32")
    np.testing.assert_allclose(out[0]["scalars"], g["scalars0"], rtol=3e-5)
    for k in g.files:
        if k.startswith("gradnorm."):
            gn = np.linalg.norm(out[0]["grads"][k[9:]].astype(np.float64))
            assert abs(gn - g[k]) <= 1e-4 * abs(g[k]) + 1e-12, k
    np.testing.assert_allclose(out[2]["scalars"], g["scalars2"], rtol=2e-4)
    sd = eng.state_dict()
    for k in g.files:
        if k.startswith("s3norm."):
            assert abs(np.linalg.norm(sd[k[7:]].astype(np.float64)) - g[k]) <= 2e-4 * abs(g[k]), k
    eng.close()


@pytest.mark.parametrize("cfgd,small,B", [(G0, True, 3), (G1, True, 4), (G1, False, 2), (G2, True, 2)])
def test_bf16_close_to_oracle(cfgd, small, B):
    """bf16 compute (the bench dtype) vs the fp32 oracle.  Stated tolerance: ELBO 2e-3 relative on
    these small random-weight nets (bf16 rounding of every stored map), gradient tensors 5e-2 rel-L2
    (weights; 0.15 on the tiny G0 net whose 8-channel layers amplify rounding through cancellation) and
    grad-norm 2e-2."""
    cfg = make_cfg(cfgd, small)
    state = init_state(cfg, 7)
    eng = E.Engine(cfg, max_batch=B, compute_dtype="bf16")
    eng.load_state(state)
    orc = vo.OracleVAE(cfg, state)
    alpha, beta = 1e6, 1e-4
    x = synthetic_samples(20251003, range(B), cfg.num_node, cfg.num_time)
    eps = synthetic_eps(1234, 0, cfg, B)
    sc, acts = engine_step(eng, cfg, x, eps, alpha, beta)
    osc, oacts, ograds = oracle_step(orc, x, eps, alpha, beta)
    le = alpha * sc["recon"] + beta * sum(sc["kls"])
    lo = alpha * osc["recon"] + beta * sum(osc["kls"])
    assert abs(le - lo) / abs(lo) < 2e-3, (le, lo)
    for k in ("enc_h0", "enc_h3", "dec_out0", "dec_out2", "x_hat"):
        assert rel_l2(acts[k], oacts[k]) < 3e-2, (k, rel_l2(acts[k], oacts[k]))
    bad = []
    for name, g in ograds.items():
        eg = eng.grad(name)
        assert (g is None) == (eg is None), name
        if g is None:
            continue
        if name.endswith("weight_orig") and rel_l2(eg, g) > (0.15 if cfgd is G0 else 5e-2):
            bad.append((name, rel_l2(eg, g)))
    assert not bad, bad
    assert abs(eng.grad_norm() - orc.grad_norm()) / orc.grad_norm() < 2e-2
    eng.close()


def test_engine_rejects_bad_config():
    cfg = make_cfg(dict(latent_dim=32, hierarchical_dim=8, enc=[30, 16, 8, 8], num_node=72, num_time=10))
    with pytest.raises(E.SgvError):
        E.Engine(cfg, max_batch=2, compute_dtype="f32")


def test_philox_eps_statistics():
    """Without injected noise the engine draws eps from its Philox stream: two forwards differ and the
    KL terms stay finite."""
    import torch
    cfg = make_cfg(G0)
    eng = E.Engine(cfg, max_batch=3, compute_dtype="f32")
    eng.load_state(init_state(cfg, 7))
    x = synthetic_samples(20251003, range(3), cfg.num_node, cfg.num_time)
    eng.set_input(torch.from_numpy(x).cuda())
    a = eng.forward(train=False)
    z1 = eng.activation("z", (3, cfg.latent_dim))
    b = eng.forward(train=False)
    z2 = eng.activation("z", (3, cfg.latent_dim))
    assert np.isfinite(a["recon"]) and np.isfinite(b["recon"])
    assert np.abs(z1 - z2).max() > 1e-3
    eng.close()


def test_engine_noise_is_keyed_by_the_global_sample_row():
    """"Philox streams keyed by global sample index so results are world-size-invariant".  Two engines that stand for
    the ranks of a 2-way data-parallel run (same seed, sgv_set_shard(r, 2), rows r::2 of the batch) must draw, row for row and
    BITWISE, the noise one engine draws for the whole batch -- at every noise site, for two consecutive steps -- and the mean of
    their gradients must be the whole-batch gradient (fp32 compute: rounding only; the sums run in a different order, so this
    part cannot be bitwise)."""
    import torch
    cfg = make_cfg(G1)
    B, T = 4, cfg.num_time
    dec = cfg.num_filter_dec
    x = synthetic_samples(20251003, range(B), cfg.num_node, cfg.num_time)
    state = init_state(cfg, 7)
    names = [e.name for e in param_spec(cfg) if e.kind in ("bias", "weight_orig", "gn_weight", "gn_bias")]

    def run(rows, rank, world):
        eng = E.Engine(cfg, max_batch=len(rows), compute_dtype="f32")
        eng.load_state(state)
        eng.seed(4711)
        eng.set_shard(rank, world)
        out = []
        for step in range(2):
            eng.set_input(torch.from_numpy(np.ascontiguousarray(x[rows])).cuda())
            eng.forward(train=True)
            n = len(rows)
            eps = [eng.activation("eps0", (n, cfg.latent_dim))] + [eng.activation(f"eps{i}", (n, T, dec[i])) for i in (1, 2)]
            eng.backward(1e6, 1e-4)
            grads = {k: eng.grad(k) for k in names}
            out.append((eps, grads))
            eng.load_state(state)          # same weights (and power-iteration vectors) for the next step: only the draw counter moves
        eng.close()
        return out
    whole = run([0, 1, 2, 3], 0, 1)
    r0, r1 = run([0, 2], 0, 2), run([1, 3], 1, 2)
    for step in range(2):
        for site in range(3):
            w = whole[step][0][site]
            assert np.array_equal(r0[step][0][site], w[[0, 2]]), (step, site)
            assert np.array_equal(r1[step][0][site], w[[1, 3]]), (step, site)
            assert np.abs(w).max() > 1.0 and not np.array_equal(w[0], w[1])
        assert not np.array_equal(whole[0][0][0], whole[1][0][0])             # a new draw every step
        for k in names:
            gw = whole[step][1][k]
            if gw is None:
                assert r0[step][1][k] is None
                continue
            mean = 0.5 * (r0[step][1][k].astype(np.float64) + r1[step][1][k].astype(np.float64))
            assert relerr(mean, gw) < 5e-5, (step, k, relerr(mean, gw))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_recompute_activations_option_regenerates_the_same_maps(dtype):
    """Option "recompute_activations" (the timing hook for BASELINE configs[3]'s "+ grad-checkpoint"): backward regenerates every
    stage's GroupNorm + GELU output from the stored pre-normalisation map before it uses it.  Same values, so the step must give
    the gradients of the plain step (fp32: bitwise -- the regenerating pass IS the forward's pass; bf16: the fused stage kernels
    normalise the same stored values, rounding only), and the hook reports how many bytes it regenerated."""
    import torch
    cfg = make_cfg(G1)
    B = 4
    x = synthetic_samples(20251003, range(B), cfg.num_node, cfg.num_time)
    eps = synthetic_eps(1234, 0, cfg, B)
    state = init_state(cfg, 7)
    names = [e.name for e in param_spec(cfg) if e.kind in ("bias", "weight_orig", "gn_weight", "gn_bias")]
    out = []
    for rec in (0, 1):
        eng = E.Engine(cfg, max_batch=B, compute_dtype=dtype)
        eng.load_state(state)
        eng.set_option("recompute_activations", rec)
        sc, _ = engine_step(eng, cfg, x, eps, 1e6, 1e-4, want_acts=False)
        out.append((sc, {k: eng.grad(k) for k in names}, eng.grad_norm(), eng.recompute_bytes()))
        eng.close()
    (sa, ga, na, ba), (sb, gb, nb, bb) = out
    assert ba == 0 and bb > 0
    assert sa == sb                                   # forward is untouched
    for k in names:
        if ga[k] is None:
            assert gb[k] is None
        elif dtype == "f32":
            assert np.array_equal(ga[k], gb[k]), k
        else:
            assert rel_l2(gb[k], ga[k]) < 2e-2, (k, rel_l2(gb[k], ga[k]))
    assert abs(na - nb) <= (0 if dtype == "f32" else 5e-3) * na


def test_augment_collate_matches_oracle():
    """A13: the fused noise/scale/mixup + collate kernel on the HBM-resident dataset vs oracle.augment_sample
    (augmentation.py:86-124).  Scale and mixup are exact arithmetic on the stored sample; the Gaussian noise comes from
    the engine's own Philox stream (the reference uses torch.randn_like), so it is checked through its statistics,
    its determinism per seed and its independence across seeds."""
    import torch
    cfg = make_cfg(G1)
    P, B = 6, 5
    x = synthetic_samples(31, range(P), cfg.num_node, cfg.num_time)
    eng = E.Engine(cfg, max_batch=B, compute_dtype="f32")
    eng.load_state(init_state(cfg, 3))
    data = torch.empty(P * eng.sample_bytes(), dtype=torch.uint8, device="cuda")
    eng.dataset_convert(torch.from_numpy(x).cuda(), data, P)
    idx = [4, 0, 5, 2, 2]
    seeds = [0, 0, 0, 12345, 777]
    scale = [1.0, 0.93, 1.07, 1.0, 1.0]
    mix = [-1, 3, 1, -1, -1]
    lam = [1.0, 0.37, 0.9, 1.0, 1.0]
    eng.augment_collate(data, idx, seeds, scale, mix, lam)
    got = eng.activation("x_in", (B, cfg.num_node, cfg.num_time))
    for b in range(3):       # no noise: exact up to one fp32 rounding per operation
        dec = dict(noise=False, scale=None if scale[b] == 1.0 else np.float32(scale[b]), lam=None if mix[b] < 0 else lam[b])
        want = vo.augment_sample(x[idx[b]], x[mix[b]] if mix[b] >= 0 else None, None, dec)
        assert relerr(got[b], want) < 1e-6, b
    for b in (3, 4):         # noise only: x + 0.05 * N(0, 1)
        nz = (got[b] - x[idx[b]]) / 0.05
        assert abs(nz.mean()) < 0.05 and abs(nz.std() - 1.0) < 0.05 and abs((nz ** 3).mean()) < 0.15
    assert abs(np.corrcoef(((got[3] - x[2]) / 0.05).ravel(), ((got[4] - x[2]) / 0.05).ravel())[0, 1]) < 0.05
    eng.augment_collate(data, idx, seeds, scale, mix, lam)
    np.testing.assert_array_equal(eng.activation("x_in", (B, cfg.num_node, cfg.num_time)), got)   # same seeds, same noise
    # ragged last batch: fewer samples than max_batch
    eng.augment_collate(data, idx[:2], seeds[:2], scale[:2], mix[:2], lam[:2])
    np.testing.assert_array_equal(eng.activation("x_in", (2, cfg.num_node, cfg.num_time)), got[:2])
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_prefetched_augmentation_is_schedule_only(dtype):
    """sgv_augment_stage / sgv_augment_advance (the training loop's prefetch: batch i + 1 is built in a second input buffer on a
    stream of its own, beside step i) against sgv_augment_collate at the start of every step: six training steps with no host
    synchronisation between them, an eval forward and a set_input in between, a staged batch that is replaced before it is
    used -- bitwise the same batches, losses and weights.  Advancing with nothing staged is an error."""
    import torch
    cfg = make_cfg(G1)
    P, B = 7, 4
    x = synthetic_samples(37, range(P), cfg.num_node, cfg.num_time)
    rs = np.random.RandomState(4)
    plans = []
    for s in range(7):
        idx = rs.randint(0, P, B).tolist()
        mix = [int((i + 1 + rs.randint(0, P - 1)) % P) if rs.rand() < 0.5 else -1 for i in idx]
        plans.append((idx, [int(rs.randint(1, 1 << 30)) if rs.rand() < 0.5 else 0 for _ in idx],
                      [float(0.9 + 0.2 * rs.rand()) for _ in idx], mix, [float(0.1 + 0.8 * rs.rand()) if m >= 0 else 1.0 for m in mix]))
    plans[5] = tuple(v[:3] for v in plans[5])              # a ragged batch in the middle of the run
    out = []
    for prefetch in (0, 1):
        eng = E.Engine(cfg, max_batch=B, compute_dtype=dtype)
        eng.load_state(init_state(cfg, 3))
        eng.seed(77)
        data = torch.empty(P * eng.sample_bytes(), dtype=torch.uint8, device="cuda")
        eng.dataset_convert(torch.from_numpy(x).cuda(), data, P)
        rec = []
        if prefetch:
            with pytest.raises(E.SgvError):
                eng.augment_advance()                       # nothing staged
            eng.augment_stage(data, *plans[6])              # replaced by the next call before it is ever used
            eng.augment_stage(data, *plans[0])
        for s in range(6):
            if prefetch:
                eng.augment_advance()
                eng.augment_stage(data, *plans[s + 1])
            else:
                eng.augment_collate(data, *plans[s])
            if s == 2:
                rec.append(eng.activation("x_in", (B, cfg.num_node, cfg.num_time)))
            eng.forward(train=True, sync=False)
            eng.backward_step(1e6, 1e-4, 1e-3)
            if s == 1:
                rec.append(eng.forward(train=False))           # same batch again, eval pass (fires nothing new: already fired)
            if s == 3:
                eng.set_input(torch.from_numpy(x[:B]).cuda())  # overwrites the CURRENT batch, not the staged one
                rec.append(eng.forward(train=False))
        rec.append(eng.forward(train=False))
        sd = eng.state_dict()
        rec.append(np.concatenate([np.asarray(sd[k], dtype=np.float32).ravel() for k in sorted(sd)]))
        if prefetch:
            eng.augment_advance()                               # plans[6], staged during the last step
            want = E.Engine(cfg, max_batch=B, compute_dtype=dtype)
            want.load_state(init_state(cfg, 3))
            want.augment_collate(data, *plans[6])
            np.testing.assert_array_equal(eng.activation("x_in", (B, cfg.num_node, cfg.num_time)),
                                          want.activation("x_in", (B, cfg.num_node, cfg.num_time)))
            want.close()
        out.append(rec)
        eng.close()
    a, b = out
    np.testing.assert_array_equal(a[1], b[1])                   # the batch of step 2
    assert a[0] == b[0] and a[2] == b[2] and a[3] == b[3]
    np.testing.assert_array_equal(a[4], b[4])


def test_api_error_behaviour():
    """Status codes + sgv_last_error text instead of crashes: wrong call order, bad sizes, unknown names."""
    import torch
    cfg = make_cfg(G0)
    eng = E.Engine(cfg, max_batch=2, compute_dtype="f32")
    with pytest.raises(E.SgvError):
        eng.backward(1.0, 1.0)                         # no forward yet
    eng.load_state(init_state(cfg, 3))
    with pytest.raises(E.SgvError):
        eng.set_input(torch.zeros((3, cfg.num_node, cfg.num_time), device="cuda"))    # batch > max_batch
    with pytest.raises(E.SgvError):
        eng.activation("enc_h0", (2, cfg.num_filter_enc[0], cfg.num_time))            # nothing computed yet
    x = torch.from_numpy(synthetic_samples(1, range(1), cfg.num_node, cfg.num_time)).cuda()
    eng.set_input(x)                                    # batch 1 of max 2
    sc = eng.forward(train=False)
    assert np.isfinite(sc["recon"])
    with pytest.raises(E.SgvError):
        eng.backward(1.0, 1.0)                         # eval forward is not differentiable state
    with pytest.raises(E.SgvError):
        eng.activation("no_such_map", (1,))
    with pytest.raises(E.SgvError):
        eng.activation("enc_h0", (1, 3))               # wrong element count
    bad = init_state(cfg, 3)
    k = next(iter(bad))
    bad[k] = np.zeros((1,), np.float32)
    with pytest.raises(E.SgvError):
        eng.load_state(bad)
    eng.forward(train=True)
    eng.backward(1.0, 1.0)
    assert eng.grad("encoder.xs_linear.0.weight_orig") is None       # dead parameter: grad=None as in the reference
    eng.close()


def test_schedule_and_fused_stage_options_agree():
    """Two engine switches on a net whose stages the fused kernels take (G2: 256-32 channels, T = 40), bf16:
    "lanes" changes the schedule only -> loss and every gradient BITWISE equal with the lane on and off;
    "fused_stages" swaps GEMM + combine + GroupNorm kernels for csrc/convgn.hip -> same step up to bf16 rounding order
    (stated: ELBO 2e-4, gradient tensors 2e-2 rel-L2)."""
    import torch
    cfg = make_cfg(G2, True)
    state = init_state(cfg, 7)
    B = 2
    x = synthetic_samples(20251003, range(B), cfg.num_node, cfg.num_time)
    eps = synthetic_eps(1234, 0, cfg, B)
    names = [k for k in state if k.endswith("weight_orig")][:24]
    out = {}
    for tag, opts in (("base", {}), ("nolanes", {"lanes": 0}), ("unfused", {"fused_stages": 0})):
        eng = E.Engine(cfg, max_batch=B, compute_dtype="bf16")
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.load_state(state)
        eng.set_input(torch.from_numpy(x).cuda())
        eng.set_eps([torch.from_numpy(e).cuda() for e in eps])
        sc = eng.forward(train=True)
        eng.backward(1e6, 1e-4)
        out[tag] = (sc, {k: eng.grad(k) for k in names if eng.grad(k) is not None}, eng.grad_norm())
        eng.close()
    sb, gb, nb = out["base"]
    sl, gl, nl = out["nolanes"]
    assert sb["recon"] == sl["recon"] and list(sb["kls"]) == list(sl["kls"]) and nb == nl
    for k in gb:
        assert np.array_equal(gb[k], gl[k]), k
    su, gu, nu = out["unfused"]
    eb = 1e6 * sb["recon"] + 1e-4 * sum(sb["kls"])
    eu = 1e6 * su["recon"] + 1e-4 * sum(su["kls"])
    assert abs(eb - eu) <= 2e-4 * abs(eu), (eb, eu)
    assert abs(nb - nu) <= 1e-2 * nu
    for k in gb:
        assert rel_l2(gb[k], gu[k]) < 2e-2, (k, rel_l2(gb[k], gu[k]))


def test_auxiliary_streams_do_not_share_the_main_streams_hardware_queue():
    """The HIP runtime maps streams onto four hardware queues and kernels of two streams on one queue never overlap; which
    queue a new stream gets depends on how many streams the process created before (this pytest process: torch's, earlier
    engines').  The engine probes every auxiliary stream at creation and rejects candidates that cannot run beside the main
    stream: second lane, side stream, optimizer stream and the engine's own communication stream must all
    report that their kernels overlap a running main-stream kernel -- for three engines created one after the other."""
    import ctypes as C
    from simulgen_vae_amd.engine import Engine
    from tests.gpu_common import G1, make_cfg
    lib = E.load_library()
    lib.sgv_test_stream_overlap.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 4:
        pytest.skip("fewer than four hardware queues configured: main / lane / side / optimizer cannot all have their own (ADVICE r2)")
    engines = [Engine(make_cfg(G1), max_batch=2, compute_dtype="bf16") for _ in range(3)]
    try:
        for eng in engines:
            for which in range(4):
                o = C.c_int(-2)
                assert lib.sgv_test_stream_overlap(eng.h, which, C.byref(o)) == 0, lib.sgv_last_error()
                assert o.value in (1, -1), (which, o.value)          # -1: the stream does not exist (option off)
                assert which > 1 or o.value == 1, (which, o.value)  # the lane and the side stream always exist
    finally:
        for eng in engines:
            eng.close()
