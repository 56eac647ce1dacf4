"""Kernel-level parity: the two MFMA GEMMs against numpy on bf16-exact inputs (so the only
difference is fp32 accumulation order).  Called through the C ABI (sgv_test_gemm_*)."""
import ctypes as C
import os

import time
import numpy as np
import pytest

import simulgen_vae_amd  # noqa: F401
from simulgen_vae_amd import engine as E

pytestmark = pytest.mark.gpu


def _bf16_round(a):
    import torch
    return torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()


def _dev(a, dtype):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(torch.bfloat16).contiguous() if dtype == 1 else t.contiguous()


def ref_conv_nt(A, W, bias, scale, addend, taps, Tlen):
    """C[m][n] = scale*sum_{j,k} A[m+j-pad][k] W[j][n][k] + bias + addend (taps masked per sample)."""
    M, K = A.shape
    N = W.shape[1]
    pad = (taps - 1) // 2
    out = np.zeros((M, N), np.float64)
    t = np.arange(M) % Tlen
    for j in range(taps):
        dt = j - pad
        src = np.arange(M) + dt
        ok = (t + dt >= 0) & (t + dt < Tlen)
        As = np.zeros_like(A, dtype=np.float64)
        As[ok] = A[src[ok]]
        out += As @ W[j].astype(np.float64).T
    out = out * scale
    if bias is not None:
        out += bias[None, :]
    if addend is not None:
        out += addend
    return out


def ref_conv_tn(dY, X, taps, Tlen):
    M, N1 = dY.shape
    N2 = X.shape[1]
    pad = (taps - 1) // 2
    out = np.zeros((taps, N1, N2), np.float64)
    t = np.arange(M) % Tlen
    for j in range(taps):
        dt = j - pad
        src = np.arange(M) + dt
        ok = (t + dt >= 0) & (t + dt < Tlen) & (src < M)      # a partial last sample: rows past M read as zeros
        Xs = np.zeros_like(X, dtype=np.float64)
        Xs[ok] = X[src[ok]]
        out[j] = dY.astype(np.float64).T @ Xs
    return out


NT_CASES = [
    # M, N, K, taps, Tlen, splitk
    (128, 128, 64, 1, 16, 1),
    (200, 72, 40, 1, 10, 1),
    (30, 8, 8, 3, 10, 1),
    (260, 136, 96, 3, 13, 1),
    (400, 200, 160, 5, 20, 1),
    (400, 200, 160, 5, 20, 3),
    (96, 264, 1032, 1, 12, 4),
    (520, 320, 320, 5, 40, 2),
    # N <= 64: the 128 x 64 form of the bf16 kernel (ragged rows, taps, split-K slabs)
    (300, 64, 96, 1, 20, 1),
    (1000, 32, 72, 3, 25, 1),
    (644, 48, 160, 5, 23, 3),
    (130, 16, 1032, 1, 13, 4),
]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", NT_CASES)
def test_gemm_nt(dtype, case):
    import torch
    lib = E.load_library()
    M, N, K, taps, Tlen, splitk = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    A = _bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    W = _bf16_round(rng.standard_normal((taps, N, K)).astype(np.float32) * 0.1)
    bias = rng.standard_normal(N).astype(np.float32)
    add = _bf16_round(rng.standard_normal((M, N)).astype(np.float32))
    scale = np.array([0.37], np.float32)
    dA, dW, dadd = _dev(A, dtype), _dev(W, dtype), _dev(add, dtype)
    dbias, dscale = torch.from_numpy(bias).cuda(), torch.from_numpy(scale).cuda()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    rc = lib.sgv_test_gemm_nt(dtype, dA.data_ptr(), dW.data_ptr(), out.data_ptr(), dbias.data_ptr(), dscale.data_ptr(),
                              dadd.data_ptr(), M, N, K, taps, Tlen, splitk, 1, None)
    assert rc == 0, lib.sgv_last_error()
    ref = ref_conv_nt(A, W, bias, 0.37, add, taps, Tlen)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err
    # compute-dtype output path (rounds to bf16 in bf16 mode)
    out2 = (torch.zeros((M, N), dtype=torch.bfloat16 if dtype == 1 else torch.float32, device="cuda"))
    rc = lib.sgv_test_gemm_nt(dtype, dA.data_ptr(), dW.data_ptr(), out2.data_ptr(), dbias.data_ptr(), None,
                              None, M, N, K, taps, Tlen, splitk, 0, None)
    assert rc == 0, lib.sgv_last_error()
    ref2 = ref_conv_nt(A, W, bias, 1.0, None, taps, Tlen)
    err2 = np.abs(out2.float().cpu().numpy() - ref2).max() / np.abs(ref2).max()
    assert err2 < (6e-3 if dtype == 1 else 2e-5), err2


NT_STATS_CASES = [
    # M, N, K, taps, Tlen, Cg, addend
    (600, 1024, 96, 1, 200, 256, False),      # sample boundaries inside tiles 1, 3, 4; group boundaries on tile edges
    (600, 1056, 64, 3, 200, 132, True),       # every 128-column tile holds a group boundary (132 = 4 * 33, not a multiple of 8)
    (520, 1360, 160, 1, 130, 170, False),     # ragged M / N, boundaries in both directions, last tiles partial
    (256, 128, 512, 1, 128, 128, False),      # exactly one tile per (sample, group)
]


@pytest.mark.parametrize("case", NT_STATS_CASES)
def test_gemm_nt_stats_epilogue(case):
    """GroupNorm statistics accumulated by the 128x128 bf16 GEMM epilogue (sgv_test_gemm_nt_stats) == per-(sample, group)
    sum and sum of squares of the bf16 output the kernel stored, in fp64; output itself against numpy.  Shapes outside
    the contract (Tlen < 128, Cg < 128, wide-kernel shapes) are rejected, not silently run without statistics."""
    import torch
    lib = E.load_library()
    M, N, K, taps, Tlen, Cg, use_add = case
    rng = np.random.default_rng(31)
    A = _bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    W = _bf16_round(rng.standard_normal((taps, N, K)).astype(np.float32) * 0.1)
    bias = rng.standard_normal(N).astype(np.float32)
    add = _bf16_round(rng.standard_normal((M, N)).astype(np.float32)) if use_add else None
    dA, dW = _dev(A, 1), _dev(W, 1)
    dadd = _dev(add, 1) if use_add else None
    dbias = torch.from_numpy(bias).cuda()
    B, G = -(-M // Tlen), N // Cg
    for rep in range(3):
        out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        sums = torch.zeros((B, G, 2), dtype=torch.float64, device="cuda")
        rc = lib.sgv_test_gemm_nt_stats(dA.data_ptr(), dW.data_ptr(), out.data_ptr(), dbias.data_ptr(),
                                        dadd.data_ptr() if use_add else None, M, N, K, taps, Tlen, Cg, sums.data_ptr(), None)
        assert rc == 0, lib.sgv_last_error()
        y = out.float().cpu().numpy().astype(np.float64)
        ref = ref_conv_nt(A, W, bias, 1.0, None, taps, Tlen)
        if use_add:
            ref = _bf16_round(ref.astype(np.float32)).astype(np.float64) + add      # the kernel rounds before adding
        assert np.abs(y - ref).max() / np.abs(ref).max() < 8e-3
        want = np.zeros((B, G, 2))
        for b in range(B):
            blk = y[b * Tlen:(b + 1) * Tlen]
            for g in range(G):
                want[b, g, 0] = blk[:, g * Cg:(g + 1) * Cg].sum()
                want[b, g, 1] = (blk[:, g * Cg:(g + 1) * Cg] ** 2).sum()
        got = sums.cpu().numpy()
        assert np.abs(got[..., 1] - want[..., 1]).max() <= 2e-6 * want[..., 1].max(), (case, rep)
        assert np.abs(got[..., 0] - want[..., 0]).max() <= 2e-6 * want[..., 1].max(), (case, rep)   # sums cancel: scale by sum sq
    dummy = torch.zeros(64, dtype=torch.float64, device="cuda")
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    assert lib.sgv_test_gemm_nt_stats(dA.data_ptr(), dW.data_ptr(), out.data_ptr(), None, None, M, N, K, taps, 100, Cg,
                                      dummy.data_ptr(), None) != 0          # Tlen < 128
    assert lib.sgv_test_gemm_nt_stats(dA.data_ptr(), dW.data_ptr(), out.data_ptr(), None, None, M, N, K, taps, Tlen, 2,
                                      dummy.data_ptr(), None) != 0          # Cg < 128


NT256_CASES = [
    # M, N, K, taps, Tlen, splitk, mode (0: the 256x256 kernel forced, 1: the engine's plan)
    (256, 256, 512, 1, 128, 1, 0),        # one tile, 8 K-tiles
    (600, 520, 200, 3, 200, 1, 0),        # ragged M / N, K tail (200 = 3*64 + 8), sample boundaries inside tiles
    (1000, 768, 1032, 5, 200, 1, 0),      # five taps, K tail of 8
    (1000, 768, 1024, 1, 200, 2, 0),      # split-K slabs + fixed-order combine
    (800, 1024, 4104, 1, 200, 3, 0),      # split-K with a K tail in the last slice
    (3200, 1280, 256, 5, 200, 1, 0),      # more work items than workgroups on an XCD label: item transitions inside a workgroup
    (256, 256, 95008, 1, 128, 4, 0),      # the first encoder layer's K (95008 = 1484*64 + 32), four slices
    (3200, 1024, 1280, 5, 200, 0, 1),     # planned: 256 kernel (all rows, or 12 row tiles + 128 tail rows on the 128-row kernel)
    (3200, 2560, 1280, 5, 200, 0, 1),     # planned: 12 x 10 tiles x 2 slices on the 256 kernel + the 128-row tail
    (3200, 1024, 512, 1, 200, 0, 1),      # planned: below the size threshold -> 128-row kernels only
    (3200, 4104, 512, 1, 200, 1, 0 | (5 << 8) | (1 << 16)),   # row tiles in bands of 5 (5 + 5 + 3), non-temporal weight stream, ragged N
    (3100, 2048, 512, 1, 200, 1, 0 | (7 << 8) | (2 << 16)),   # bands of 7 + 6, ragged last row tile, sc1 output stores
    (3200, 1280, 320, 3, 200, 2, 0 | (4 << 8)),               # bands with taps and split-K slices
    (3200, 16640, 512, 1, 200, 1, 0),                         # the launcher's own choice for a one-tap product with >= 64 column panels
    # 128 x 512 tiles (bit 19): all eight waves share the 128 rows, 80 KiB K-tile buffers, vmcnt(10)
    (256, 512, 512, 1, 128, 1, 0 | (1 << 19)),               # two row tiles of one column tile
    (3200, 1024, 1032, 5, 200, 1, 0 | (1 << 19)),            # 25 row tiles x 2 column tiles, five taps, K tail of 8
    (3200, 1536, 4104, 1, 200, 3, 0 | (1 << 19)),            # split-K slabs with a K tail in the last slice
    (3000, 1288, 520, 3, 200, 1, 0 | (1 << 19)),             # ragged last row / column tile, sample boundaries inside the tiles
    (3200, 5120, 640, 3, 200, 1, 0 | (1 << 19)),             # 250 items: one round of the chip
    (3200, 10240, 256, 5, 200, 1, 0 | (1 << 19) | (5 << 8)), # 500 items, row tiles in bands of 5: item transitions inside a workgroup
    (256, 512, 95008, 1, 128, 4, 0 | (1 << 19)),             # the first encoder layer's K, four slices
    (3200, 2560, 1280, 5, 200, 0, 1 | (2 << 19)),            # planned with the 128 x 512 tile forbidden: main + tail as in round 2
    # bit 21: the planned launch with its 128-row tail as a launch of its own on shifted row pointers (the engine runs it BESIDE the
    # main launch on the lane stream): one round of <= 16 items of the 128 x 512 tile, GemmNT::trow0 = 3072 % 200 = 72 for the tap windows
    (3200, 1024, 8200, 1, 200, 3, 1 | (1 << 21)),            # one tap, K tail of 8, main launch in 3 slices, tail in 5
    (3200, 2560, 1280, 5, 200, 2, 1 | (1 << 21)),            # five taps: the tail's first rows continue the sample that starts in the main rows
    (3200, 2560, 1536, 3, 128, 1, 1 | (1 << 21)),            # three taps, sample length 128: the tail starts exactly on a sample boundary
    (1408, 1288, 520, 3, 88, 1, 1 | (1 << 21)),              # ragged column tile, sample length 88 (1280 % 88 = 48)
]


@pytest.mark.parametrize("case", NT256_CASES)
def test_gemm_nt256(case):
    """256x256 persistent LDS-DMA kernel (csrc/gemm256.hip) against numpy on bf16-exact inputs: fp32 output (2e-5, fp32
    accumulation order only) and bf16 output with scale, bias and addend; forced and through the engine's kernel plan."""
    import torch
    lib = E.load_library()
    M, N, K, taps, Tlen, splitk, mode = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    A = _bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    W = _bf16_round(rng.standard_normal((taps, N, K)).astype(np.float32) * 0.1)
    bias = rng.standard_normal(N).astype(np.float32)
    add = _bf16_round(rng.standard_normal((M, N)).astype(np.float32))
    dA, dW, dadd = _dev(A, 1), _dev(W, 1), _dev(add, 1)
    dbias, dscale = torch.from_numpy(bias).cuda(), torch.tensor([0.37], device="cuda")
    kind = C.c_int(-1)
    ref = ref_conv_nt(A, W, bias, 0.37, None, taps, Tlen)
    base = mode & 0xff  # bits 8+: work-item order / weight-stream policy (include/sgvae.h)
    if base == 0:       # fp32 output: final (split-K 1) or slabs + combine
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        rc = lib.sgv_test_gemm_nt256(dA.data_ptr(), dW.data_ptr(), out.data_ptr(), dbias.data_ptr(), dscale.data_ptr(), None, M, N, K,
                                     taps, Tlen, splitk, 1, mode, 0, None, C.byref(kind), None)
        assert rc == 0, lib.sgv_last_error()
        got = out.cpu().numpy()
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-5
    out2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc = lib.sgv_test_gemm_nt256(dA.data_ptr(), dW.data_ptr(), out2.data_ptr(), dbias.data_ptr(), dscale.data_ptr(), dadd.data_ptr(), M, N, K,
                                 taps, Tlen, splitk, 0, mode, 0, None, C.byref(kind), None)
    assert rc == 0, lib.sgv_last_error()
    if (mode >> 21) & 1:
        assert kind.value == 2, kind.value                  # forced: 256-row tiles + the 128-row tail on shifted pointers
    elif base == 1:
        if (mode >> 19) & 3 == 2:
            assert kind.value != 3, kind.value
        assert (kind.value in (1, 2, 3)) if N * K * taps >= 1024 * 1280 * 5 else kind.value == 0, kind.value
    want = _bf16_round(ref.astype(np.float32)).astype(np.float64) + add        # the kernels round before adding the addend
    got2 = out2.float().cpu().numpy()
    assert np.isfinite(got2).all()
    assert np.abs(got2 - want).max() / np.abs(want).max() < 8e-3


NT256_STATS_CASES = [
    # M, N, K, taps, Tlen, Cg
    (600, 1024, 512, 1, 200, 256),       # sample boundaries inside the waves' 128-row halves; group boundaries on wave edges
    (600, 1056, 512, 3, 200, 132),       # 132 = 4 * 33: group boundaries inside 64-column blocks and inside 8-column chunks
    (520, 1360, 520, 1, 130, 68),        # ragged M / N, Cg just above the 64-column limit, K tail
    (3200, 2048, 512, 1, 200, 256),      # several items per workgroup
    (3200, 2048, 512, 1, 200, 256, 4),   # the same with the row tiles in bands of 4: the statistics partials are indexed by tile, not by list position
    (3000, 4224, 512, 1, 200, 132, 6),   # bands of 6 + 6, ragged last row tile, groups inside column blocks
    (3200, 2048, 512, 1, 200, 256, 0, 1),   # 128 x 512 tiles: wave w = 128 rows x 64 columns at (0, 64 w)
    (3000, 4224, 520, 1, 200, 132, 9, 1),   # the same with bands of 9 row tiles, ragged tiles, groups inside column blocks, K tail
]


@pytest.mark.parametrize("case", NT256_STATS_CASES)
def test_gemm_nt256_stats_epilogue(case):
    """GroupNorm statistics from the 256x256 kernel's epilogue (per-(item, wave) partial sums + fixed-order finalize, no atomics)
    == per-(sample, group) sum / sum of squares of the bf16 output the kernel stored (fp64), and bitwise equal between runs."""
    import torch
    lib = E.load_library()
    M, N, K, taps, Tlen, Cg = case[:6]
    order = (case[6] << 8) | (3 << 16) if len(case) > 6 else 0     # bands + non-temporal weights + sc1 output
    if len(case) > 7 and case[7]:
        order = ((case[6] or 255) << 8) | (1 << 19)                # 128 x 512 tiles (no stream policies there)
    rng = np.random.default_rng(37)
    A = _bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    W = _bf16_round(rng.standard_normal((taps, N, K)).astype(np.float32) * 0.1)
    bias = rng.standard_normal(N).astype(np.float32)
    dA, dW, dbias = _dev(A, 1), _dev(W, 1), torch.from_numpy(bias).cuda()
    B, G = -(-M // Tlen), N // Cg
    ref = ref_conv_nt(A, W, bias, 1.0, None, taps, Tlen)
    first = None
    for rep in range(2):
        out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        sums = torch.full((B, G, 2), float("nan"), dtype=torch.float64, device="cuda")      # the finalize overwrites: no zero-fill needed
        rc = lib.sgv_test_gemm_nt256(dA.data_ptr(), dW.data_ptr(), out.data_ptr(), dbias.data_ptr(), None, None, M, N, K, taps, Tlen, 1, 0,
                                     order, Cg, sums.data_ptr(), None, None)
        assert rc == 0, lib.sgv_last_error()
        y = out.float().cpu().numpy().astype(np.float64)
        assert np.abs(y - ref).max() / np.abs(ref).max() < 8e-3
        want = np.zeros((B, G, 2))
        for b in range(B):
            blk = y[b * Tlen:(b + 1) * Tlen]
            for g in range(G):
                want[b, g, 0] = blk[:, g * Cg:(g + 1) * Cg].sum()
                want[b, g, 1] = (blk[:, g * Cg:(g + 1) * Cg] ** 2).sum()
        got = sums.cpu().numpy()
        assert np.abs(got[..., 1] - want[..., 1]).max() <= 2e-6 * want[..., 1].max(), (case, rep)
        assert np.abs(got[..., 0] - want[..., 0]).max() <= 2e-6 * want[..., 1].max(), (case, rep)
        if first is None:
            first = got.copy()
        else:
            assert np.array_equal(first, got)          # deterministic: bitwise equal between launches
    dummy = torch.zeros(64, dtype=torch.float64, device="cuda")
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    assert lib.sgv_test_gemm_nt256(dA.data_ptr(), dW.data_ptr(), out.data_ptr(), None, None, None, M, N, K, taps, 100, 1, 0, 0, Cg,
                                   dummy.data_ptr(), None, None) != 0          # Tlen < 128


TN_CASES = [
    # M, N1, N2, taps, Tlen, splitk
    (128, 128, 128, 1, 16, 1),
    (200, 72, 40, 1, 10, 1),
    (30, 8, 8, 3, 10, 1),
    (260, 136, 96, 3, 13, 2),
    (400, 200, 160, 5, 20, 1),
    (400, 200, 160, 5, 20, 3),
    (1040, 264, 328, 1, 40, 4),
]


@pytest.mark.parametrize("dtype,use_tr", [(0, 0), (1, 1), (1, 0)])
@pytest.mark.parametrize("case", TN_CASES)
def test_gemm_tn(dtype, use_tr, case):
    import torch
    lib = E.load_library()
    M, N1, N2, taps, Tlen, splitk = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    dY = _bf16_round(rng.standard_normal((M, N1)).astype(np.float32))
    X = _bf16_round(rng.standard_normal((M, N2)).astype(np.float32))
    ddY, dX = _dev(dY, dtype), _dev(X, dtype)
    out = torch.full((taps, N1, N2), float("nan"), dtype=torch.float32, device="cuda")
    rc = lib.sgv_test_gemm_tn(dtype, ddY.data_ptr(), dX.data_ptr(), out.data_ptr(), M, N1, N2, taps, Tlen, splitk,
                              use_tr, None)
    assert rc == 0, lib.sgv_last_error()
    ref = ref_conv_tn(dY, X, taps, Tlen)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err


TN_W2_CASES = [
    # M, N1, N2, taps, Tlen, splitk  -- all eligible for the 128x256 LDS-DMA weight-gradient kernel (bf16, N2 >= 256, Tlen >= 32)
    (1600, 192, 520, 5, 200, 1),      # ragged N1/N2 tiles, taps across sample boundaries
    (1000, 136, 256, 3, 100, 3),      # M not a multiple of 32, split-K slabs
    (3200, 1024, 768, 1, 200, 2),
    (777, 64, 264, 5, 64, 1),         # M not a multiple of Tlen
    (4000, 320, 1280, 3, 200, 5),
    (512, 128, 256, 1, 512, 1),
    (640, 72, 296, 5, 32, 1),         # smallest allowed Tlen: every 32-row stage starts on a sample boundary
    (990, 128, 256, 5, 33, 2),        # Tlen = 33: the boundary walks through every row of the stage
    (1360, 200, 512, 3, 34, 1),
    (288, 64, 256, 5, 48, 1),         # 9 stages
    (400, 128, 95008, 1, 200, 1),     # the first encoder layer's width on the input side (95008 = 371 * 256 + 32)
    (400, 95008, 256, 1, 200, 1),     # the recon head's width on the output side (95008 = 742 * 128 + 32): 743 items on 512 persistent blocks
    (640, 1408, 2560, 5, 64, 1),      # 11 x 10 tiles x 5 taps = 550 items: patches of tiles x taps, several items per block
    (1280, 1288, 1288, 3, 128, 4),    # 11 x 6 tiles x 3 taps x 4 slices = 792 items, ragged tiles
]


@pytest.mark.parametrize("persistent", [0, 1])
@pytest.mark.parametrize("case", TN_W2_CASES)
def test_gemm_tn_w2(case, persistent):
    """gemm_tn_w2_kernel (128x256 tile, 32-row stages, two blocks per CU) vs numpy on bf16-exact inputs, 8x per shape, with one
    work item per block and with the persistent walk over patches of tiles x taps (what the launcher picks for the big multi-tap
    gradients); both orders must give bitwise the same matrix."""
    import torch
    lib = E.load_library()
    M, N1, N2, taps, Tlen, splitk = case
    rng = np.random.default_rng(13)
    dY = _bf16_round(rng.standard_normal((M, N1)).astype(np.float32))
    X = _bf16_round(rng.standard_normal((M, N2)).astype(np.float32))
    ddY, dX = _dev(dY, 1), _dev(X, 1)
    ref = ref_conv_tn(dY, X, taps, Tlen)
    scale = np.abs(ref).max()
    for rep in range(8):
        out = torch.full((taps, N1, N2), float("nan"), dtype=torch.float32, device="cuda")
        # use_tr = 2: force the two-blocks-per-CU kernel for every eligible shape (whatever SGV_TN_W2 says); 3: and its persistent walk
        rc = lib.sgv_test_gemm_tn(1, ddY.data_ptr(), dX.data_ptr(), out.data_ptr(), M, N1, N2, taps, Tlen, splitk, 2 + persistent, None)
        assert rc == 0, lib.sgv_last_error()
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), (case, rep)
        err = np.abs(got - ref).max() / scale
        assert err < 2e-5, (case, rep, err)
    if persistent:
        other = torch.full((taps, N1, N2), float("nan"), dtype=torch.float32, device="cuda")
        assert lib.sgv_test_gemm_tn(1, ddY.data_ptr(), dX.data_ptr(), other.data_ptr(), M, N1, N2, taps, Tlen, splitk, 2, None) == 0
        assert torch.equal(out, other)          # the item order changes which block computes a tile, not a single sum


TN256_CASES = [
    # M, N1, N2, taps, Tlen, splitk -- eligible for the 256 x 256 weight-gradient kernel (bf16, M % 64 == 0, M >= 1024, widths >= 256, Tlen >= 64)
    (1024, 256, 256, 1, 128, 1),       # one tile, 16 K-tiles
    (1280, 520, 776, 3, 64, 1),        # ragged tiles, three taps, sample boundaries on K-tile edges
    (3200, 512, 768, 5, 200, 1),       # five taps, sample boundaries inside the K-tiles and inside the DMA pieces
    (3200, 1024, 512, 1, 200, 2),      # split-K slabs (25 K-tiles each)
    (1600, 256, 95008, 1, 200, 1),     # the first encoder layer's width: 372 items on 256 workgroups, item transitions
    (1600, 95008, 256, 1, 200, 1),     # the recon head's width on the row side
    (2048, 768, 1280, 5, 128, 2),      # taps x slices
    (1280, 2560, 2816, 5, 64, 1),      # 10 x 11 tiles x 5 taps = 550 items: several items per workgroup with taps
    (3264, 264, 328, 3, 96, 1),        # M = 34 samples of 96: windows that neither start nor end on a K-tile edge
]


@pytest.mark.parametrize("case", TN256_CASES)
def test_gemm_tn256(case):
    """gemm_tn_t256_kernel (csrc/gemm256tn.hip: 256 x 256 tiles, persistent, transposed LDS reads, taps as work items) vs numpy on
    bf16-exact inputs, 4x per shape (the counted-vmcnt / barrier pipeline must not race), and bitwise equal between launches."""
    import torch
    lib = E.load_library()
    M, N1, N2, taps, Tlen, splitk = case
    rng = np.random.default_rng(23)
    dY = _bf16_round(rng.standard_normal((M, N1)).astype(np.float32))
    X = _bf16_round(rng.standard_normal((M, N2)).astype(np.float32))
    ddY, dX = _dev(dY, 1), _dev(X, 1)
    ref = ref_conv_tn(dY, X, taps, Tlen)
    scale = np.abs(ref).max()
    first = None
    for rep in range(4):
        out = torch.full((taps, N1, N2), float("nan"), dtype=torch.float32, device="cuda")
        rc = lib.sgv_test_gemm_tn(1, ddY.data_ptr(), dX.data_ptr(), out.data_ptr(), M, N1, N2, taps, Tlen, splitk, 4, None)   # 4: force the 256 x 256 kernel
        assert rc == 0, lib.sgv_last_error()
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), (case, rep)
        err = np.abs(got - ref).max() / scale
        assert err < 2e-5, (case, rep, err)
        if first is None:
            first = got
        else:
            assert np.array_equal(first, got), (case, rep)
    # shapes the kernel does not take are refused by the forced hook's launcher and fall to the 128 x 256 kernel: still correct
    if M % 128 == 0:
        out = torch.full((taps, N1, N2), float("nan"), dtype=torch.float32, device="cuda")
        Mh = M - 32            # not a multiple of 64
        assert lib.sgv_test_gemm_tn(1, ddY.data_ptr(), dX.data_ptr(), out.data_ptr(), Mh, N1, N2, taps, Tlen, 1, 4, None) == 0
        r2 = ref_conv_tn(dY[:Mh], X[:Mh], taps, Tlen)
        assert np.abs(out.cpu().numpy() - r2).max() / np.abs(r2).max() < 2e-5


@pytest.mark.parametrize("case", [c for c in TN256_CASES if c[5] == 1])
def test_gemm_tn256_bf16_output(case):
    """The same kernel with its bf16 epilogue (GemmTN::out_bf16, engine option grad_bf16): exactly the fp32 result rounded to nearest
    even -- the accumulators are the same, only the store differs -- including the ragged edge tiles (no store outside the matrix: the
    canary behind it survives)."""
    import torch
    lib = E.load_library()
    M, N1, N2, taps, Tlen, _ = case
    rng = np.random.default_rng(29)
    dY = _bf16_round(rng.standard_normal((M, N1)).astype(np.float32))
    X = _bf16_round(rng.standard_normal((M, N2)).astype(np.float32))
    ddY, dX = _dev(dY, 1), _dev(X, 1)
    ref32 = torch.full((taps, N1, N2), float("nan"), dtype=torch.float32, device="cuda")
    assert lib.sgv_test_gemm_tn(1, ddY.data_ptr(), dX.data_ptr(), ref32.data_ptr(), M, N1, N2, taps, Tlen, 1, 4, None) == 0, lib.sgv_last_error()
    n = taps * N1 * N2
    out = torch.full((n + 64,), -7.0, dtype=torch.bfloat16, device="cuda")
    assert lib.sgv_test_gemm_tn(1, ddY.data_ptr(), dX.data_ptr(), out.data_ptr(), M, N1, N2, taps, Tlen, 1, 6, None) == 0, lib.sgv_last_error()
    assert torch.equal(out[:n].view(taps, N1, N2), ref32.to(torch.bfloat16))
    assert bool((out[n:] == -7.0).all())


@pytest.mark.parametrize("case", [TN256_CASES[0], TN256_CASES[2], TN256_CASES[3], TN256_CASES[4], TN256_CASES[7], (3200, 2560, 2560, 5, 200, 1)])
@pytest.mark.parametrize("occupied", [0, 48])
def test_gemm_tn256_work_stealing(case, occupied):
    """The work-stealing form of the 256 x 256 weight-gradient kernel (GemmTN::sched; what a data-parallel backward launches while a
    collective's channel workgroups may hold CUs): bitwise the static kernel's result -- on a free chip (nobody is late: nothing is
    taken) and with `occupied` workgroups of 512 threads and 64 KiB of LDS spinning on another stream for 3 ms, so that as many of
    the kernel's workgroups are placed late, find their lists taken and join the thieves.  Shapes: one item (seven of eight
    workgroups have no list), 90 and 16 items (grid < 256), 372 items, split-K, 550 and 500 items with taps."""
    import torch
    lib = E.load_library()
    M, N1, N2, taps, Tlen, splitk = case
    rng = np.random.default_rng(31)
    dY = _bf16_round(rng.standard_normal((M, N1)).astype(np.float32))
    X = _bf16_round(rng.standard_normal((M, N2)).astype(np.float32))
    ddY, dX = _dev(dY, 1), _dev(X, 1)
    ref = torch.full((taps, N1, N2), float("nan"), dtype=torch.float32, device="cuda")
    assert lib.sgv_test_gemm_tn(1, ddY.data_ptr(), dX.data_ptr(), ref.data_ptr(), M, N1, N2, taps, Tlen, splitk, 4, None) == 0, lib.sgv_last_error()
    side = torch.cuda.Stream()
    for rep in range(3):
        out = torch.full((taps, N1, N2), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        if occupied:
            assert lib.sgv_test_occupy(side.cuda_stream, occupied, 512, 65536, 300000) == 0, lib.sgv_last_error()
            time.sleep(0.0005)
        assert lib.sgv_test_gemm_tn(1, ddY.data_ptr(), dX.data_ptr(), out.data_ptr(), M, N1, N2, taps, Tlen, splitk, 7, None) == 0, lib.sgv_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (case, occupied, rep)


@pytest.mark.parametrize("case", [(520, 320, 1024, 5, 40, 2), (640, 512, 2048, 1, 128, 1), (400, 1024, 640, 5, 200, 1),
                                  (300, 296, 9008, 1, 100, 2), (256, 512, 4160, 1, 64, 1)])
def test_gemm_nt_wide_stress(case):
    """The 128x256 LDS-DMA kernel (bf16, N >= 256, >= 64 K-steps) repeated 25x per shape: every run must match
    (guards the counted-vmcnt / barrier pipeline against intermittent races).  The last two shapes are the long one-tap
    contraction of the K = 95 008 layers in small: ragged M / N, a K tail, 1 and 2 split-K slices."""
    import torch
    lib = E.load_library()
    M, N, K, taps, Tlen, splitk = case
    rng = np.random.default_rng(7)
    A = _bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    W = _bf16_round(rng.standard_normal((taps, N, K)).astype(np.float32) * 0.05)
    dA, dW = _dev(A, 1), _dev(W, 1)
    ref = ref_conv_nt(A, W, None, 1.0, None, taps, Tlen)
    scale = np.abs(ref).max()
    for it in range(25):
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        rc = lib.sgv_test_gemm_nt(1, dA.data_ptr(), dW.data_ptr(), out.data_ptr(), None, None, None, M, N, K, taps, Tlen,
                                  splitk, 1, None)
        assert rc == 0, lib.sgv_last_error()
        err = np.abs(out.cpu().numpy() - ref).max() / scale
        assert err < 2e-5, (it, err)


@pytest.mark.parametrize("case", [(3200, 5120, 1024, True, True, False), (3200, 1280, 512, True, False, True),
                                  (3200, 512, 2560, False, True, True), (1600, 2560, 512, False, False, False),
                                  (3200, 128, 256, True, True, False)])
def test_gemm_nt_library_path(case):
    """The hand-written NT kernels against hipBLASLt on the same operands.  The library is a COMPARATOR only (round 3 took its
    dispatch out of libsgvae.so: tests/micro/vendor/libsgvcmp.so, built by __graft_entry__.build()): scale from a device scalar
    (handed over as a vector), fp32 bias along N, optional bf16 residual addend, bf16 output -- both against numpy, and against
    each other (one fp32 result rounded to bf16 on both sides: equal up to summation order, a few one-ulp flips)."""
    import ctypes
    cmp_path = os.path.join(os.path.dirname(__file__), "micro", "vendor", "libsgvcmp.so")
    if not os.path.exists(cmp_path):
        pytest.skip("comparator library not built (make -C tests/micro/vendor)")
    cmp_lib = ctypes.CDLL(cmp_path)
    vp_ = ctypes.c_void_p
    cmp_lib.sgv_cmp_gemm_nt_lib.argtypes = [vp_, vp_, vp_, vp_, vp_, vp_, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp_]
    import torch
    lib = E.load_library()
    M, N, K, use_bias, use_scale, use_add = case
    rng = np.random.default_rng(17)
    A = _bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    W = _bf16_round(rng.standard_normal((1, N, K)).astype(np.float32) * 0.05)
    bias = rng.standard_normal(N).astype(np.float32)
    add = _bf16_round(rng.standard_normal((M, N)).astype(np.float32))
    dA, dW, dadd = _dev(A, 1), _dev(W, 1), _dev(add, 1)
    dbias = torch.from_numpy(bias).cuda() if use_bias else None
    dscale = torch.tensor([0.37], dtype=torch.float32, device="cuda") if use_scale else None
    args = (dbias.data_ptr() if use_bias else None, dscale.data_ptr() if use_scale else None, dadd.data_ptr() if use_add else None)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    rc = cmp_lib.sgv_cmp_gemm_nt_lib(dA.data_ptr(), dW.data_ptr(), out.data_ptr(), *args, M, N, K, None)
    if rc == 2:
        pytest.skip("libhipblaslt.so.1 not loadable or no algorithm for this shape")
    assert rc == 0, rc
    ref = ref_conv_nt(A, W, bias if use_bias else None, 0.37 if use_scale else 1.0, add if use_add else None, 1, M)
    got = out.float().cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-3
    own = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    rc = lib.sgv_test_gemm_nt(1, dA.data_ptr(), dW.data_ptr(), own.data_ptr(), *args, M, N, K, 1, M, 1, 0, None)
    assert rc == 0, lib.sgv_last_error()
    d = np.abs(got - own.float().cpu().numpy())
    if use_add:     # the own epilogue rounds to bf16 before adding the residual and again after; the library rounds once
        assert d.max() / np.abs(ref).max() < 1.6e-2 and d.mean() / np.abs(ref).mean() < 3e-3
    else:
        assert d.max() / np.abs(ref).max() < 8e-3 and d.mean() / np.abs(ref).mean() < 1e-4
    # shapes round 1 kept on its own kernels are refused by the comparator
    assert cmp_lib.sgv_cmp_gemm_nt_lib(dA.data_ptr(), dW.data_ptr(), out.data_ptr(), None, None, None, 64, 64, 64, None) == 1


CONV_GN_CASES = [
    # B, T, N, K, taps, G, residual
    (2, 200, 512, 512, 3, 8, False),       # preset decoder block: Cg = 64, 13 row tiles (the last one with 8 valid rows)
    (3, 200, 1024, 1024, 3, 8, True),      # widest fused layer: Cg = 128, residual add
    (2, 200, 128, 256, 1, 8, False),       # Cg = 16, one tap
    (2, 40, 256, 64, 5, 8, True),          # Cg = 32, five taps, short samples (3 row tiles: one wave idles)
    (1, 16, 128, 32, 3, 8, False),         # a single row tile, a single K chunk
    (2, 208, 256, 96, 3, 8, False),        # T at the kernel's maximum, K = 3 chunks
    (2, 200, 640, 128, 1, 8, False),       # Cg = 80 (five column tiles): the 5C-channel decoder residual layers
    (2, 200, 1280, 256, 1, 8, True),       # Cg = 160 (ten column tiles; the column owners of the backward kernel span three waves)
]


@pytest.mark.parametrize("case", CONV_GN_CASES)
def test_conv_gn_fused_forward(case):
    """csrc/convgn.hip (one launch: convolution + GroupNorm + GELU [+ residual]) against numpy in fp64 on bf16-exact inputs:
    stored pre-norm output y (one bf16 rounding of the exact value), group sums of the stored y, the activated output, and a
    bitwise-equal second launch."""
    import math
    import torch
    lib = E.load_library()
    B, T, N, K, taps, G, use_res = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    A = _bf16_round(rng.standard_normal((B * T, K)).astype(np.float32))
    W = _bf16_round((rng.standard_normal((taps, N, K)) / math.sqrt(K * taps)).astype(np.float32))
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    gamma = (1.0 + 0.2 * rng.standard_normal(N)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(N)).astype(np.float32)
    res = _bf16_round(rng.standard_normal((B * T, N)).astype(np.float32))
    scale = np.array([0.83], np.float32)
    y64 = ref_conv_nt(A, W, bias, 0.83, None, taps, T)
    y_ref = _bf16_round(y64.astype(np.float32)).astype(np.float64)                 # what the kernel stores and normalises
    Cg = N // G
    yg = y_ref.reshape(B, T, G, Cg)
    s1 = yg.sum(axis=(1, 3)); s2 = (yg ** 2).sum(axis=(1, 3))
    mean = s1 / (T * Cg); var = np.maximum(s2 / (T * Cg) - mean ** 2, 0.0)
    z = (yg - mean[:, None, :, None]) / np.sqrt(var[:, None, :, None] + 1e-5)
    z = z.reshape(B * T, N) * gamma[None, :] + beta[None, :]
    erf = np.vectorize(math.erf)
    act = 0.5 * z * (1.0 + erf(z / math.sqrt(2.0)))
    want = res + 0.1 * act if use_res else act
    dA, dW, dres = _dev(A, 1), _dev(W, 1), _dev(res, 1)
    dbias, dgamma, dbeta, dscale = (torch.from_numpy(v).cuda() for v in (bias, gamma, beta, scale))
    outs = []
    for rep in range(2):
        y = torch.full((B * T, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        out = torch.full((B * T, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        sums = torch.full((B, G, 2), float("nan"), dtype=torch.float64, device="cuda")
        rc = lib.sgv_test_conv_gn_fwd(dA.data_ptr(), dW.data_ptr(), dbias.data_ptr(), dscale.data_ptr(), dres.data_ptr() if use_res else None,
                                      dgamma.data_ptr(), dbeta.data_ptr(), y.data_ptr(), out.data_ptr(), sums.data_ptr(), B, T, N, K, taps, G,
                                      C.c_float(0.1 if use_res else 1.0), None)
        assert rc == 0, lib.sgv_last_error()
        outs.append((y.float().cpu().numpy(), out.float().cpu().numpy(), sums.cpu().numpy()))
    yk, ok, sk = outs[0]
    assert np.isfinite(yk).all() and np.isfinite(ok).all() and np.isfinite(sk).all()
    # y: fp32 accumulation in another order may flip a bf16 rounding: one bf16 ulp (2^-8 relative) on few entries
    assert np.abs(yk - y_ref).max() <= 2 ** -7 * np.abs(y_ref).max()
    assert np.mean(np.abs(yk - y_ref)) <= 2e-4 * np.mean(np.abs(y_ref))
    # sums of the values the kernel itself stored: exact up to fp32 partial sums
    ykg = yk.astype(np.float64).reshape(B, T, G, Cg)
    np.testing.assert_allclose(sk[..., 0], ykg.sum(axis=(1, 3)), rtol=0, atol=2e-4 * np.abs(ykg).sum(axis=(1, 3)).max())
    np.testing.assert_allclose(sk[..., 1], (ykg ** 2).sum(axis=(1, 3)), rtol=2e-5)
    assert np.abs(ok - want).max() <= 2e-2 * max(1.0, np.abs(want).max())
    assert np.mean(np.abs(ok - want)) <= 3e-3 * np.mean(np.abs(want))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)                                  # deterministic
    # shapes the kernel does not take are refused (the engine then uses the GEMM + GroupNorm kernels)
    assert lib.sgv_test_conv_gn_fwd(dA.data_ptr(), dW.data_ptr(), dbias.data_ptr(), None, None, dgamma.data_ptr(), dbeta.data_ptr(),
                                    y.data_ptr(), out.data_ptr(), sums.data_ptr(), B, T, N, K, taps, 3, C.c_float(1.0), None) != 0


@pytest.mark.parametrize("case", CONV_GN_CASES)
def test_conv_gn_fused_backward(case):
    """csrc/convgn.hip backward mirror (input gradient of the upper convolution [+ the residual path's addend] + GroupNorm / GELU
    backward of the stage below in one launch) against numpy in fp64: dY, the group sums (s1, s2), the per-sample column totals and the <G, W_eff> partials;
    a second launch is bitwise equal."""
    import math
    import torch
    lib = E.load_library()
    B, T, N, K, taps, G, use_res = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 1)
    dYup = _bf16_round((0.5 * rng.standard_normal((B * T, K))).astype(np.float32))
    W = _bf16_round((rng.standard_normal((taps, N, K)) / math.sqrt(K * taps)).astype(np.float32))
    y = _bf16_round(rng.standard_normal((B * T, N)).astype(np.float32))
    gamma = (1.0 + 0.2 * rng.standard_normal(N)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(N)).astype(np.float32)
    cbias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    scale = np.array([0.83], np.float32)
    Cg = N // G
    cnt = T * Cg
    add = _bf16_round((0.3 * rng.standard_normal((B * T, N))).astype(np.float32)) if use_res else None     # residual path of the block above
    dA = _bf16_round(ref_conv_nt(dYup, W, None, 0.83, add, taps, T).astype(np.float32)).astype(np.float64)
    use_pre = taps == 3                     # the upper convolution reads GELU(x): gradient times gelu'(x), rounded again
    xpre = _bf16_round(rng.standard_normal((B * T, N)).astype(np.float32)) if use_pre else None
    if use_pre:
        xp = xpre.astype(np.float64)
        gp_ = 0.5 * (1.0 + np.vectorize(math.erf)(xp / math.sqrt(2.0))) + xp * np.exp(-0.5 * xp * xp) / math.sqrt(2.0 * math.pi)
        dA = _bf16_round((dA * gp_).astype(np.float32)).astype(np.float64)
    yg = y.astype(np.float64).reshape(B, T, G, Cg)
    S = yg.sum(axis=(1, 3)); SS = (yg ** 2).sum(axis=(1, 3))
    mean = S / cnt; var = np.maximum(SS / cnt - mean ** 2, 0.0); rstd = 1.0 / np.sqrt(var + 1e-5)
    xh = (yg - mean[:, None, :, None]) * rstd[:, None, :, None]
    gm = gamma.astype(np.float64).reshape(G, Cg)[None, None]; bt = beta.astype(np.float64).reshape(G, Cg)[None, None]
    z = xh * gm + bt
    erf = np.vectorize(math.erf)
    gprime = 0.5 * (1.0 + erf(z / math.sqrt(2.0))) + z * np.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)
    dz = dA.reshape(B, T, G, Cg) * gprime
    s1 = (gm * dz).sum(axis=(1, 3)); s2 = (gm * dz * xh).sum(axis=(1, 3))
    dy_ref = rstd[:, None, :, None] * (gm * dz - s1[:, None, :, None] / cnt - xh * s2[:, None, :, None] / cnt)
    colA = dz.sum(axis=1); colB = (dz * xh).sum(axis=1); colX = xh.sum(axis=1)                      # [B][G][Cg]
    colD = rstd[:, :, None] * (gm[0] * colA - T * s1[:, :, None] / cnt - s2[:, :, None] / cnt * colX)
    dot_ref = (dy_ref * (yg - cbias.astype(np.float64).reshape(G, Cg)[None, None])).sum(axis=(1, 3))
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dA_dev, dW, dy_in = _dev(dYup, 1), _dev(W, 1), _dev(y, 1)
    dadd = _dev(add, 1) if use_res else None
    dpre = _dev(xpre, 1) if use_pre else None
    sums = d(np.stack([S, SS], axis=-1))
    dg, db, dcb, dsc = d(gamma), d(beta), d(cbias), d(scale)
    outs = []
    for rep in range(2):
        dy = torch.full((B * T, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        sums2 = torch.full((B, G, 2), float("nan"), dtype=torch.float64, device="cuda")
        ptot = torch.full((B, 3, N), float("nan"), dtype=torch.float32, device="cuda")
        cdot = torch.full((B, G), float("nan"), dtype=torch.float32, device="cuda")
        da_out = torch.full((B * T, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        rc = lib.sgv_test_conv_gn_bwd(dA_dev.data_ptr(), dW.data_ptr(), dsc.data_ptr(), dadd.data_ptr() if use_res else None,
                                      dpre.data_ptr() if use_pre else None, da_out.data_ptr(), dy_in.data_ptr(), sums.data_ptr(), dg.data_ptr(),
                                      db.data_ptr(), dcb.data_ptr(), dy.data_ptr(), sums2.data_ptr(), ptot.data_ptr(), cdot.data_ptr(),
                                      B, T, N, K, taps, G, None)
        assert rc == 0, lib.sgv_last_error()
        outs.append((dy.float().cpu().numpy(), sums2.cpu().numpy(), ptot.cpu().numpy(), cdot.cpu().numpy(), da_out.float().cpu().numpy()))
    dyk, s2k, ptk, cdk, dak = outs[0]
    # the stored input gradient: one or two bf16 roundings of the exact value
    assert np.abs(dak - dA).max() <= 2 ** -6 * np.abs(dA).max() and np.mean(np.abs(dak - dA)) <= 1e-3 * np.mean(np.abs(dA))
    for a in outs[0]:
        assert np.isfinite(a).all()
    dyr = dy_ref.reshape(B * T, N)
    assert np.abs(dyk - dyr).max() <= 2e-2 * np.abs(dyr).max()
    assert np.mean(np.abs(dyk - dyr)) <= 4e-3 * np.mean(np.abs(dyr))
    scale12 = np.abs(gm * dz).sum(axis=(1, 3))                      # s1 / s2 are sums with cancellation: scale by the sum of magnitudes
    assert np.abs(s2k[..., 0] - s1).max() <= 3e-3 * scale12.max()
    assert np.abs(s2k[..., 1] - s2).max() <= 3e-3 * np.abs(gm * dz * xh).sum(axis=(1, 3)).max()
    pt = ptk.reshape(B, 3, G, Cg)
    assert np.abs(pt[:, 0] - colA).max() <= 3e-3 * np.abs(dz).sum(axis=1).max()
    assert np.abs(pt[:, 1] - colB).max() <= 3e-3 * np.abs(dz * xh).sum(axis=1).max()
    assert np.abs(pt[:, 2] - colD).max() <= 5e-3 * (np.abs(colD).max() + np.abs(rstd[:, :, None] * gm[0] * np.abs(dz).sum(axis=1)).max())
    assert np.abs(cdk - dot_ref).max() <= 5e-3 * np.abs(dy_ref * (yg - cbias.reshape(G, Cg)[None, None])).sum(axis=(1, 3)).max()
    for a, b2 in zip(outs[0], outs[1]):
        assert np.array_equal(a, b2)
