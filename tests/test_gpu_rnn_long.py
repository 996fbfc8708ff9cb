"""Recurrent parity at utterance length.  The reference trains its (Bi)LSTM / (Bi)GRU groups on whole
utterances (rnn_dyn/RNNWrapper.py:89-102: torch.nn.LSTM / GRU on a PackedSequence), i.e. fp32
gradients flow through ~1 300 .. 2 000 recurrent steps; SURVEY.md section 7 asks for a per-tensor
tolerance budget at that length.  Here: one bidirectional layer, in 425, H 512, ragged batches of 8
and 17 rows with T = 2 000 and 1 300 frames, forward and backward against torch.nn.LSTM / GRU in
float64 (torch's own per-step ATen implementation; it runs on the GPU as well -- MIOpen has no
float64 RNN -- which takes seconds where the host cores take minutes).

Budget (asserted below; measured values in DESIGN.md section 11): outputs and final states
5e-6 absolute on O(1) values; every gradient tensor 2e-5 of its largest entry.  Measured on MI355X:
outputs 5e-7 .. 9e-7, final states 3e-7 .. 6e-7, input gradient 2e-6 .. 3e-6, weight and bias
gradients 6e-7 .. 2.1e-6 -- the same level as at T = 44 (tests/test_gpu_rnn_config3.py): nothing
drifts with the sequence length.  The recurrent weight gradient is a sum over T x B outer products
accumulated by the split-K slabs of the fp32 GEMM in a fixed order; its rounding error grows like
sqrt(T B) * 6e-8 relative."""
import json
import os

import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

pytestmark = pytest.mark.gpu

OUT_ABS = 5e-6
GRAD_REL = 2e-5


def _lengths(B, T, seed):
    rng = np.random.default_rng(seed)
    lens = rng.integers(T // 2, T + 1, size=B)
    lens[0] = T
    lens[-1] = T // 3
    return torch.from_numpy(lens[rng.permutation(B)].astype(np.int64))


@pytest.mark.parametrize("cell,B,T", [("LSTM", 8, 2000), ("LSTM", 17, 1300), ("GRU", 8, 2000), ("GRU", 17, 1300)])
def test_long_sequences_match_torch_float64(gpu, cell, B, T):
    from idiaptts_amd import nn as inn
    in_dim, H = 425, 512
    torch.manual_seed(1000 + B + T)
    mine = getattr(inn, cell)(in_dim, H, 1, bidirectional=True).to(gpu)
    ref = getattr(torch.nn, cell)(in_dim, H, 1, bidirectional=True).double().to(gpu)
    ref.load_state_dict({k: v.detach().double() for k, v in mine.state_dict().items()})
    lens = _lengths(B, T, B + T)
    x = torch.randn(T, B, in_dim)
    for b, l in enumerate(lens.tolist()):
        x[l:, b] = 3.0
    w = torch.randn(T, B, 2 * H) / np.sqrt(T)      # keeps the gradients O(1)
    xr = x.double().to(gpu).requires_grad_(True)
    out_p, hn_ref = ref(pack_padded_sequence(xr, lens, enforce_sorted=False))
    out_ref, _ = pad_packed_sequence(out_p, total_length=T)
    (out_ref * w.double().to(gpu)).sum().backward()
    out_ref, hn_ref = out_ref.cpu(), (hn_ref.cpu() if cell == "GRU" else (hn_ref[0].cpu(), hn_ref[1].cpu()))
    xr_grad = xr.grad.cpu()

    xg = x.to(gpu).requires_grad_(True)
    out, hn = mine(xg, None, lens)
    (out * w.to(gpu)).sum().backward()
    torch.cuda.synchronize()

    report = {"cell": cell, "B": B, "T": T}
    report["out_abs"] = (out.detach().cpu().double() - out_ref.detach()).abs().max().item()
    hn_m = hn if cell == "GRU" else hn[0]
    hn_r = hn_ref if cell == "GRU" else hn_ref[0]
    report["hn_abs"] = (hn_m.cpu().double() - hn_r.detach()).abs().max().item()
    report["dx_rel"] = ((xg.grad.cpu().double() - xr_grad).abs().max() / xr_grad.abs().max()).item()
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        report["d" + n] = ((pm.grad.cpu().double() - pr.grad.cpu()).abs().max() / pr.grad.abs().max().cpu()).item()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "rnn_long_%s_%d_%d.json" % (cell, B, T)), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    assert report["out_abs"] < OUT_ABS and report["hn_abs"] < OUT_ABS, report
    for k, v in report.items():
        if k.startswith("d"):
            assert v < GRAD_REL, (k, v, report)


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_config3_bench_shape_matches_torch_float64(gpu, cell):
    """BASELINE config 3 at the EXACT shape bench.py times (`bilstm` / `bigru`): 3 x 512 bidirectional
    layers, 425 inputs, the 64 utterances of make_ff_batch(64, seed=7) (T = 1977, 73 138 valid
    frames): four 16-row tiles per direction on all eight XCDs, ~2 000 steps of the polling exchange,
    forward and backward, against torch.nn.LSTM / GRU in float64 on the GPU.  Same budget as above."""
    from idiaptts_amd import nn as inn
    from idiaptts_amd.bench_support import utterance_lengths
    in_dim, H, layers = 425, 512, 3
    lens = torch.from_numpy(utterance_lengths(64, seed=7))
    B, T = 64, int(lens.max())
    assert T == 1977 and int(lens.sum()) == 73138
    torch.manual_seed(64 + T)
    mine = getattr(inn, cell)(in_dim, H, layers, bidirectional=True).to(gpu)
    ref = getattr(torch.nn, cell)(in_dim, H, layers, bidirectional=True).double().to(gpu)
    ref.load_state_dict({k: v.detach().double() for k, v in mine.state_dict().items()})
    x = torch.randn(T, B, in_dim)
    for b, l in enumerate(lens.tolist()):
        x[l:, b] = 3.0
    w = torch.randn(T, B, 2 * H) / np.sqrt(T)
    xr = x.double().to(gpu).requires_grad_(True)
    out_p, hn_ref = ref(pack_padded_sequence(xr, lens, enforce_sorted=False))
    out_ref, _ = pad_packed_sequence(out_p, total_length=T)
    (out_ref * w.double().to(gpu)).sum().backward()
    out_ref = out_ref.detach().cpu()
    hn_r = (hn_ref if cell == "GRU" else hn_ref[0]).detach().cpu()
    xr_grad = xr.grad.cpu()

    xg = x.to(gpu).requires_grad_(True)
    out, hn = mine(xg, None, lens)
    (out * w.to(gpu)).sum().backward()
    torch.cuda.synchronize()
    report = {"cell": cell, "B": B, "T": T, "layers": layers}
    report["out_abs"] = (out.detach().cpu().double() - out_ref).abs().max().item()
    report["hn_abs"] = ((hn if cell == "GRU" else hn[0]).cpu().double() - hn_r).abs().max().item()
    report["dx_rel"] = ((xg.grad.cpu().double() - xr_grad).abs().max() / xr_grad.abs().max()).item()
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        report["d" + n] = ((pm.grad.cpu().double() - pr.grad.cpu()).abs().max() / pr.grad.abs().max().cpu()).item()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "rnn_config3_%s.json" % cell), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    assert report["out_abs"] < OUT_ABS and report["hn_abs"] < OUT_ABS, report
    for k, v in report.items():
        if k.startswith("d"):
            assert v < GRAD_REL, (k, v, report)
