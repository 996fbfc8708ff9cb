"""GPU parity of the dense acoustic-model kernels (through the C ABI) against plain torch fp32
on the CPU -- i.e. against the very calls the reference makes (torch.nn.Linear + Tanh:
rnn_dyn/FFWrapper.py:63-73; MSELoss*mask 'mean_per_frame': loss/NamedLoss.py:70-117;
torch.optim.Adam: ModularModelHandlerPyTorch.py:570-571). Tolerances are stated per test."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).norm().item() / (b.norm().item() + 1e-30)


@pytest.mark.parametrize("M,N,K,act", [
    (1, 1, 1, 0), (5, 7, 3, 1), (128, 128, 32, 0), (130, 187, 512, 0), (257, 512, 425, 1),
    (1000, 512, 512, 2), (333, 67, 409, 2), (64, 32, 36, 1),
])
def test_linear_fwd(gpu, M, N, K, act):
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(M * 31 + N)
    x = torch.rand(M, K, generator=g)
    w = (torch.rand(N, K, generator=g) - 0.5) * (2.0 / K ** 0.5)
    b = torch.rand(N, generator=g) - 0.5
    z = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = [z, torch.tanh(z), torch.relu(z)][act].float()
    y = ops.linear_fwd(x.to(gpu), w.to(gpu), b.to(gpu), act).cpu()
    # fp32 MFMA is an exact k-ordered fmaf chain: error ~1e-7*sum|a b|; allow 2e-6 relative
    assert _rel(y, ref) < 2e-6
    assert (y - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("M,N,K", [(3000, 2500, 1024), (700, 4096, 1028), (1500, 1100, 2048)])
def test_linear_fwd_with_weights_larger_than_the_l2_walks_column_groups(gpu, M, N, K):
    """Weight matrices above ~4 MB change the tile order of the forward product (groups of column
    tiles, nn.hip ring_group / ring::decode_tile); N is chosen so that the last group is narrower.
    Every output element is checked (reference: torch fp64 on the GPU)."""
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.rand(M, K, generator=g).to(gpu)
    w = ((torch.rand(N, K, generator=g) - 0.5) * (2.0 / K ** 0.5)).to(gpu)
    b = (torch.rand(N, generator=g) - 0.5).to(gpu)
    ref = torch.tanh(torch.nn.functional.linear(x.double(), w.double(), b.double()))
    y = ops.linear_fwd(x, w, b, 1).double()
    assert _rel(y, ref) < 2e-6
    assert (y - ref).abs().max().item() < 2e-5


def test_linear_fwd_strided_rows(gpu):
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(3)
    big = torch.rand(300, 440, generator=g)
    x = big[:, 5:430]          # K = 425, row stride 440, unaligned base
    w = torch.rand(187, 425, generator=g) - 0.5
    y = ops.linear_fwd(x.to(gpu)[:, :], w.to(gpu), None, 0)
    ref = x.double() @ w.double().t()
    assert _rel(y.cpu().double(), ref) < 2e-6


@pytest.mark.parametrize("M,N,K,actp", [(5, 7, 3, 1), (300, 187, 512, 1), (1000, 512, 425, 0),
                                        (129, 512, 512, 2)])
def test_linear_bwd_input(gpu, M, N, K, actp):
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(M + N)
    dz = torch.randn(M, N, generator=g)
    w = torch.randn(N, K, generator=g) / N ** 0.5
    yprev = torch.tanh(torch.randn(M, K, generator=g))
    ref = dz.double() @ w.double()
    if actp == 1:
        ref = ref * (1 - yprev.double() ** 2)
    elif actp == 2:
        ref = ref * (yprev.double() > 0)
    dx = ops.linear_bwd_input(dz.to(gpu), w.to(gpu), yprev.to(gpu) if actp else None, actp).cpu()
    assert _rel(dx.double(), ref) < 2e-6


@pytest.mark.parametrize("M,N,K", [(5, 7, 3), (300, 187, 512), (5000, 512, 425), (70000, 64, 96),
                                   (1, 4, 4)])
def test_linear_bwd_weight(gpu, M, N, K):
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(M + K)
    dz = torch.randn(M, N, generator=g)
    x = torch.rand(M, K, generator=g)
    dw, db = ops.linear_bwd_weight(dz.to(gpu), x.to(gpu))
    rw = dz.double().t() @ x.double()
    rb = dz.double().sum(0)
    assert _rel(dw.cpu().double(), rw) < 3e-6
    assert _rel(db.cpu().double(), rb) < 3e-6
    # accumulate adds on top
    dw2, db2 = ops.linear_bwd_weight(dz.to(gpu), x.to(gpu), dw=dw.clone(), db=db.clone(),
                                     accumulate=True)
    assert _rel(dw2.cpu().double(), 2 * rw) < 3e-6
    assert _rel(db2.cpu().double(), 2 * rb) < 3e-6
    # determinism (slab split-K, no float atomics): bit-identical on a re-run
    dw3, db3 = ops.linear_bwd_weight(dz.to(gpu), x.to(gpu))
    assert torch.equal(dw3, dw) and torch.equal(db3, db)


def test_masked_mse_matches_namedloss(gpu):
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, D = 4, 50, 187
    lengths = torch.tensor([50, 31, 1, 44])
    pred = torch.randn(B, T, D, generator=g, requires_grad=True)
    target = torch.randn(B, T, D, generator=g)
    mask = (torch.arange(T)[None, :] < lengths[:, None]).unsqueeze(-1).float()
    # NamedLoss._reduce 'mean_per_frame' (loss/NamedLoss.py:113-117)
    v = torch.nn.MSELoss(reduction='none')(target, pred) * mask
    ref = (v.sum(dim=(0, 1)) / lengths.sum().float()).mean()
    ref.backward()
    valid = (mask.reshape(-1) > 0).to(torch.uint8)
    loss, grad = ops.masked_mse(pred.detach().reshape(-1, D).to(gpu), target.reshape(-1, D).to(gpu),
                                valid.to(gpu), float(lengths.sum()))
    assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
    assert _rel(grad.cpu(), pred.grad.reshape(-1, D)) < 1e-6
    assert (grad.cpu()[valid == 0] == 0).all()
    loss2, g2 = ops.masked_mse(pred.detach().reshape(-1, D).to(gpu), target.reshape(-1, D).to(gpu),
                               valid.to(gpu), float(lengths.sum()), want_grad=False)
    assert g2 is None and loss2.item() == loss.item()


def test_adam_matches_torch(gpu):
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(10007, generator=g)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    p = p0.clone().to(gpu)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step in range(1, 6):
        grad = torch.randn(10007, generator=g)
        p_ref.grad = grad.clone()
        opt.step()
        ops.adam_step(p, grad.to(gpu), m, v, step, lr=1e-3)
        assert (p.cpu() - p_ref.detach()).abs().max().item() < 2e-7


def test_ff_train_steps_match_reference_stack(gpu):
    """3 Adam steps of the 425-512-512-187 model on packed frames == the reference's padded
    torch-CPU step (same seeds, fp32). Tolerance: 1e-5 relative on the loss, 2e-6 abs on
    parameters after 3 steps of lr 1e-3 (Adam normalises the update, so grads that agree to
    ~1e-6 relative move a weight identically up to ~lr*1e-3)."""
    from idiaptts_amd.bench_support import TorchRefFF, make_ff_batch, pad_batch, torch_ref_step
    from idiaptts_amd.native_ff import FlatFFModel
    dims, acts = (425, 512, 512, 187), ("tanh", "tanh", None)
    model = FlatFFModel(dims, acts, device=gpu, seed=4)
    ref = TorchRefFF([(w.cpu(), b.cpu()) for w, b in model.layers()], acts)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    for step in range(3):
        x, y, lengths = make_ff_batch(3, seed=100 + step)
        lt = torch.from_numpy(lengths)
        ref_loss = torch_ref_step(ref, opt, pad_batch(x, lt), pad_batch(y, lt), lt)
        valid = torch.ones(x.shape[0], dtype=torch.uint8, device=gpu)
        loss = model.train_step(x.to(gpu), y.to(gpu), valid, float(lengths.sum()))
        assert abs(loss.item() - ref_loss.item()) < 1e-5 * abs(ref_loss.item())
    lins = [m for m in ref.net if isinstance(m, torch.nn.Linear)]
    for i, lin in enumerate(lins):
        assert (model.weight(i).cpu() - lin.weight.detach()).abs().max().item() < 2e-6
        assert (model.bias(i).cpu() - lin.bias.detach()).abs().max().item() < 2e-6


def test_degenerate_sizes_and_error_reporting(gpu):
    """Edge cases through the C ABI: empty batches are no-ops (or zero the gradients), bad
    arguments come back as IttsError with the library's message (no crash, no silent fallback)."""
    from idiaptts_amd import lib, ops
    w = torch.randn(5, 8, device=gpu)
    b = torch.randn(5, device=gpu)
    empty = torch.empty((0, 8), device=gpu)
    assert ops.linear_fwd(empty, w, b, 1).shape == (0, 5)
    dz = torch.empty((0, 5), device=gpu)
    assert ops.linear_bwd_input(dz, w).shape == (0, 8)
    dw, db = ops.linear_bwd_weight(dz, empty)
    assert float(dw.abs().sum()) == 0.0 and float(db.abs().sum()) == 0.0
    # one frame, one column
    y = ops.linear_fwd(torch.ones((1, 1), device=gpu), torch.full((1, 1), 2.0, device=gpu),
                       torch.full((1,), 0.5, device=gpu), 0)
    assert float(y) == 2.5
    with pytest.raises(ValueError):
        ops.linear_fwd(torch.randn(3, 7, device=gpu), w, b)             # K mismatch
    with pytest.raises(lib.IttsError) as e:
        ops.mlpg_generation(torch.zeros((4, 2), dtype=torch.float64, device=gpu),
                            torch.ones(3, dtype=torch.float64, device=gpu), 1, [0, 4])
    assert "leading dimension" in str(e.value)
    from idiaptts_amd.nn.functional import PackedBatch
    with pytest.raises(ValueError):
        PackedBatch([3, 5], 4, False, gpu)                                # length beyond the padding
    # recurrent entry points refuse unsorted lengths instead of computing garbage
    L = lib.load()
    import ctypes
    hl = (ctypes.c_int * 2)(2, 3)
    dummy = torch.zeros(64, device=gpu)
    rc = L.itts_lstm_layer_fwd(ctypes.c_void_p(dummy.data_ptr()), ctypes.c_void_p(dummy.data_ptr()),
                               None, None, ctypes.c_void_p(dummy.data_ptr()), hl,
                               ctypes.c_void_p(dummy.data_ptr()), ctypes.c_void_p(dummy.data_ptr()),
                               3, 2, 16, 1, ctypes.c_void_p(dummy.data_ptr()), None, None, None, None,
                               ctypes.c_void_p(dummy.data_ptr()), None)
    assert rc != 0 and b"longest" in L.itts_last_error()


@pytest.mark.parametrize("type_,reduction,masked,batch_first", [
    ("MSELoss", "mean_per_sample", True, False), ("MSELoss", "mean", True, False),
    ("MSELoss", "sum", True, True), ("MSELoss", "none", True, False),
    ("MSELoss", "mean", False, False), ("MSELoss", "sum", False, True),
    ("L1Loss", "mean_per_frame", True, False), ("L1Loss", "mean_per_sample", True, True),
    ("L1Loss", "mean", False, False), ("L1Loss", "none", True, False)])
def test_named_loss_other_types_and_reductions(gpu, type_, reduction, masked, batch_first):
    """NamedLoss.forward / _reduce (reference loss/NamedLoss.py:70-131) for the combinations the
    hot path does not use: loss_fn(reduction='none') * seq_mask, then the reduction -- value and
    gradient against the same formulas in torch on the CPU (float64)."""
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler
    from idiaptts_amd.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    torch.manual_seed(3)
    lens = torch.tensor([7, 3, 5])
    T, B, D = 7, 3, 4
    shape = (B, T, D) if batch_first else (T, B, D)
    pred = torch.randn(shape)
    target = torch.randn(shape)
    mask = Handler.sequence_mask(lens, T, batch_first=batch_first)
    loss_mod = NamedLoss.Config(name="l", type_=type_, seq_mask="m" if masked else None,
                                input_names=["y", "p"], batch_first=batch_first,
                                reduction=reduction, loss_weight=0.7).create_loss()
    pg = pred.to(gpu).requires_grad_(True)
    data = {"y": target.to(gpu), "p": pg, "m": mask.to(gpu)}
    out = loss_mod(data, {"m": lens, "y": lens, "p": lens}, step=1)["l"]
    w_out = torch.randn(out.shape).to(gpu) if reduction == "none" else None
    (out * w_out).sum().backward() if reduction == "none" else out.backward()
    # reference formulas
    pr = pred.double().requires_grad_(True)
    v = getattr(torch.nn, type_)(reduction="none")(target.double(), pr)
    if masked:
        v = mask.double() * v
    if reduction == "mean_per_frame":
        ref = (v.sum(dim=(0, 1)) / lens.sum().double()).mean()
    elif reduction == "mean_per_sample":
        ref = (v.sum(dim=1 if batch_first else 0) / lens.unsqueeze(-1).double()).mean()
    elif reduction == "mean":
        ref = v.mean()
    elif reduction == "sum":
        ref = v.sum()
    else:
        ref = v
    ref = ref * 0.7
    (ref * w_out.cpu().double()).sum().backward() if reduction == "none" else ref.backward()
    assert out.shape == ref.shape
    assert (out.detach().cpu().double() - ref.detach()).abs().max() < 1e-5 * max(1.0, float(ref.abs().max()))
    assert (pg.grad.cpu().double() - pr.grad).abs().max() < 1e-5 * max(1.0, float(pr.grad.abs().max()))


def test_output_layer_fused_with_the_masked_mse_equals_the_two_kernels(gpu):
    """itts_linear_fwd_mse (the output layer's GEMM epilogue forms the loss and its gradient; the
    layer's output is never stored) against itts_linear_fwd + itts_masked_mse, on ragged sizes incl.
    an output width that is no multiple of 4 and masked rows."""
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(5)
    for M, K, N in [(1000, 512, 187), (129, 64, 4), (4097, 128, 62), (77, 32, 187)]:
        x = torch.randn(M, K, generator=g).to(gpu)
        w = (torch.randn(N, K, generator=g) * 0.1).to(gpu)
        b = torch.randn(N, generator=g).to(gpu)
        pitch = (N + 3) // 4 * 4
        target = torch.randn(M, N, generator=g).to(gpu)
        valid = (torch.rand(M, generator=g) > 0.2).to(torch.uint8).to(gpu)
        n_valid = float(valid.sum().item())
        y = ops.linear_fwd(x, w, b, ops.ACT_NONE)
        loss_ref, dz_ref = ops.masked_mse(y, target, valid, n_valid)
        dz = torch.zeros(M, pitch, device=gpu)[:, :N]
        loss, dz = ops.linear_fwd_mse(x, w, b, target, valid, n_valid, grad=dz)
        assert abs(float(loss) - float(loss_ref)) <= 1e-6 * abs(float(loss_ref))
        assert torch.equal(dz, dz_ref)          # same float arithmetic per element
        # and against torch in fp64
        yd = x.double().cpu() @ w.double().cpu().T + b.double().cpu()
        d = (yd - target.double().cpu()) * valid.cpu().double()[:, None]
        assert abs(float(loss) - float((d ** 2).sum() / (n_valid * N))) < 1e-5 * float(loss_ref)


@pytest.mark.parametrize("M,N,K", [(3000, 512, 428), (700, 64, 96), (300, 188, 64)])
def test_weight_and_bias_gradients_in_one_flat_arena(gpu, M, N, K):
    """db right behind dw in one buffer (the layout of the flat gradient arenas): the bias sums come
    out of the weight-gradient GEMM and one slab reduction writes both; also on top of what the
    arena already holds (accumulate)."""
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(M + N)
    dz = torch.randn(M, N, generator=g)
    x = torch.rand(M, K, generator=g)
    rw = dz.double().t() @ x.double()
    rb = dz.double().sum(0)
    flat = torch.zeros(N * K + N, device=gpu)
    dw, db = flat[:N * K].view(N, K), flat[N * K:]
    ops.linear_bwd_weight(dz.to(gpu), x.to(gpu), dw=dw, db=db)
    assert _rel(dw.cpu().double(), rw) < 3e-6 and _rel(db.cpu().double(), rb) < 3e-6
    first = flat.clone()
    ops.linear_bwd_weight(dz.to(gpu), x.to(gpu), dw=dw, db=db, accumulate=True)
    assert _rel(dw.cpu().double(), 2 * rw) < 3e-6 and _rel(db.cpu().double(), 2 * rb) < 3e-6
    # the separate-buffer form gives the same bits
    dw2, db2 = ops.linear_bwd_weight(dz.to(gpu), x.to(gpu))
    assert torch.equal(dw2.reshape(-1), first[:N * K]) and torch.equal(db2, first[N * K:])


@pytest.mark.parametrize("M,N,K,with_prev", [(3001, 187, 512, True), (2500, 512, 512, True), (777, 64, 96, False),
                                             (130, 187, 425, True)])
def test_fused_backward_equals_the_separate_calls_bit_for_bit(gpu, M, N, K, with_prev):
    """itts_linear_bwd (weight + bias + input gradient of a layer in one launch) against
    itts_linear_bwd_weight + itts_linear_bwd_input and against torch in float64."""
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(M + N)
    Kp = (K + 3) // 4 * 4
    x = torch.zeros(M, Kp)
    x[:, :K] = torch.tanh(torch.randn(M, K, generator=g))
    dz = torch.randn(M, N, generator=g)
    w = torch.zeros(N, Kp)
    w[:, :K] = torch.randn(N, K, generator=g) * 0.1
    xg, wg = x.to(gpu), w.to(gpu)
    Np = (N + 3) // 4 * 4
    dzg = torch.zeros(M, Np, device=gpu)[:, :N]
    dzg.copy_(dz)
    flat = torch.zeros(N * Kp + Np, device=gpu)       # db right behind dw: the merged slab reduction
    dw, db = flat[:N * Kp].view(N, Kp), flat[N * Kp:N * Kp + N]
    dx = torch.zeros(M, Kp, device=gpu)
    yprev = xg if with_prev else None
    ops.linear_bwd(dzg, xg, wg, dw, db, dx, yprev=yprev, act_prev=ops.ACT_TANH if with_prev else ops.ACT_NONE)
    flat2 = torch.zeros_like(flat)
    dw2, db2 = flat2[:N * Kp].view(N, Kp), flat2[N * Kp:N * Kp + N]
    ops.linear_bwd_weight(dzg, xg, dw=dw2, db=db2)
    dx2 = ops.linear_bwd_input(dzg, wg, yprev=yprev, act_prev=ops.ACT_TANH if with_prev else ops.ACT_NONE,
                               out=torch.zeros(M, Kp, device=gpu))
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dx, dx2)
    dw_ref = dz.double().t() @ x.double()
    dx_ref = dz.double() @ w.double()
    if with_prev:
        dx_ref = dx_ref * (1 - x.double() ** 2)
    assert (dw.cpu().double() - dw_ref).abs().max() < 2e-4 * max(1.0, dw_ref.abs().max().item())
    assert (db.cpu().double() - dz.double().sum(0)).abs().max() < 2e-4 * max(1.0, dz.double().sum(0).abs().max().item())
    assert (dx.cpu().double() - dx_ref).abs().max() < 1e-5 * max(1.0, dx_ref.abs().max().item())


def test_deferred_reductions_equal_the_separate_launches_bit_for_bit(gpu):
    """ops.deferred_reductions(): loss partial sums, merged weight + bias slabs, a separate odd-sized
    bias vector (187) and more reductions than one launch holds (17 > 16), all reduced at the end of
    the block -- every result bit-equal to the immediate launches; and the dense stack's
    loss_and_backward (which defers) against its undeferred body."""
    from idiaptts_amd import ops
    from idiaptts_amd.native_ff import FlatFFModel
    g = torch.Generator().manual_seed(5)

    def one_layer(M, N, K, defer):
        x = torch.tanh(torch.randn(M, K, generator=torch.Generator().manual_seed(M))).to(gpu)
        dz = torch.zeros(M, (N + 3) // 4 * 4, device=gpu)[:, :N]
        dz.copy_(torch.randn(M, N, generator=torch.Generator().manual_seed(N)))
        dw = torch.zeros(N, K, device=gpu)
        db = torch.zeros(N, device=gpu)
        ops.linear_bwd_weight(dz, x, dw=dw, db=db)
        return dw, db

    shapes = [(3001, 187, 512), (2500, 512, 512)] + [(400 + 37 * i, 64, 96) for i in range(15)]
    ref = [one_layer(*s, defer=False) for s in shapes]
    with ops.deferred_reductions():
        got = [one_layer(*s, defer=True) for s in shapes]
    torch.cuda.synchronize()
    for (dw, db), (dw2, db2) in zip(ref, got):
        assert torch.equal(dw, dw2) and torch.equal(db, db2)

    model = FlatFFModel([425, 512, 512, 187], ["tanh", "tanh", None], device=gpu, seed=3)
    M = 2001
    x = torch.randn(M, 425, generator=g).to(gpu)
    y = torch.randn(M, 187, generator=g).to(gpu)
    valid = (torch.rand(M, generator=g) > 0.1).to(torch.uint8).to(gpu)
    nv = float(valid.sum().item())
    loss_d = model.loss_and_backward(x, y, valid, nv).clone()
    grads_d = model.grads.clone()
    model.grads.zero_()
    loss_i = model._loss_and_backward(model.pack_input(x), y, valid, nv, None, False).clone()
    torch.cuda.synchronize()
    assert torch.equal(loss_d, loss_i) and torch.equal(grads_d, model.grads)
    assert float(loss_d) > 0


def test_gemm_operands_that_end_where_their_mapping_ends(gpu):
    """Column slices of a [4096 x 128] float tensor that is its own 2 MiB allocation as operands of
    the weight-gradient GEMM (what the recurrent layers pass: dG[:, d * 4H:], h_prev[:, d * H:]): the
    ring kernel's operand descriptors must end at the last valid element -- ending at the last
    row's pitch they reached past the allocation by the slice's column offset and the launch died
    with a memory access fault.  Child process: the allocator has to be switched to one hipMalloc per
    tensor before torch starts, and a regression kills the process."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts",
                          "desc_end_of_mapping.py")
    env = dict(os.environ, PYTORCH_NO_CUDA_MEMORY_CACHING="1")
    res = subprocess.run([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_rows_gather_is_pack_unpack_and_shift(gpu):
    """itts_rows_gather_f32 behind PackedBatch (pack_padded_sequence / pad_packed_sequence of
    rnn_dyn/RNNWrapper.py:89-102 and their gradients): against torch's own packing, padded widths,
    fill rows, column slices as operands, and the gradient through pack -> unpack."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from idiaptts_amd import ops
    from idiaptts_amd.nn.functional import PackedBatch
    torch.manual_seed(0)
    T, B, F = 23, 5, 425
    lens = torch.tensor([23, 7, 15, 1, 19])
    x = torch.randn(T, B, F, device=gpu, requires_grad=True)
    pb = PackedBatch(lens, T, False, gpu)
    packed = pb.pack(x, pad_cols=True)
    assert packed.shape == (int(lens.sum()), 428) and bool((packed[:, 425:] == 0).all())
    ref = pack_padded_sequence(x.detach().cpu(), lens, enforce_sorted=False)
    assert torch.equal(packed[:, :F].detach().cpu(), ref.data)
    out = pb.unpack(packed[:, :F] * 2.0, x.shape)
    want, _ = pad_packed_sequence(torch.nn.utils.rnn.PackedSequence(ref.data * 2.0, ref.batch_sizes, ref.sorted_indices,
                                                                     ref.unsorted_indices), total_length=T)
    assert torch.equal(out.detach().cpu(), want)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    mask = (torch.arange(T)[:, None] < lens[None, :]).to(gpu)[..., None]
    assert torch.equal(x.grad, (2.0 * w * mask))
    # plain gather with a fill row, out-of-range indices and a column slice as the source
    src = torch.randn(10, 64, device=gpu)
    idx = torch.tensor([3, 10, 0, -1, 9], device=gpu)
    fill = torch.randn(16, device=gpu)
    got = ops.rows_gather(src[:, 16:32], idx, fill_row=fill, out_width=20)
    want = torch.zeros(5, 20, device=gpu)
    for r, i in enumerate(idx.tolist()):
        want[r, :16] = src[i, 16:32] if 0 <= i < 10 else fill
    assert torch.equal(got, want)
