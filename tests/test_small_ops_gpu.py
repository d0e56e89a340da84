"""GPU parity of the GEMM / GRU / small kernels against fp32 torch on the CPU.
Reference layers: nn.Linear / nn.GRU of two_d_cnn_lstm (model/baseline_models.py:191-210),
cloak_noise (cloak_models.py:24-58), train() loss (training_cloak_with_grl.py:143-160),
optimisers (:416-421)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(5600, 192, 1280), (224, 128, 128), (224, 4, 128), (200, 192, 2048), (37, 53, 71),
                                   (64, 64, 32), (1, 2, 3)])
@pytest.mark.parametrize("abf16", [False, True])
def test_linear_three_products(M, N, K, abf16):
    from sept_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    if abf16:
        x = x.bfloat16()
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xf = x.float()
    y = ops.linear_forward(x.cuda(), W.cuda(), b.cuda()).cpu()
    assert torch.allclose(y, xf @ W.t() + b, rtol=1e-4, atol=1e-4)
    dx = ops.linear_backward_input(dy.cuda(), W.cuda()).cpu()
    assert torch.allclose(dx, dy @ W, rtol=1e-4, atol=1e-4)
    dW = ops.linear_backward_weight(dy.cuda(), x.cuda()).cpu()      # split-K path when K(=M) is long
    want = dy.t() @ xf
    assert torch.allclose(dW, want, rtol=1e-4, atol=1e-4 * want.abs().max())
    assert torch.equal(dW, ops.linear_backward_weight(dy.cuda(), x.cuda()).cpu())   # deterministic
    assert torch.allclose(ops.colsum(dy.cuda()).cpu(), dy.sum(0), rtol=1e-4, atol=1e-4)
    # strided views + bf16 output + beta accumulate (the GRU dx path)
    out = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    ops.gemm_raw(dy.cuda(), N, 1, W.cuda(), K, 1, out, K, M, K, N)
    ops.gemm_raw(dy.cuda(), N, 1, W.cuda(), K, 1, out, K, M, K, N, beta=1.0)
    assert torch.allclose(out.float().cpu(), 2 * (dy @ W), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("M,N,K,abf16", [(5600, 384, 1280, True), (5600, 1280, 384, False), (70, 100, 96, True),
                                         (33, 65, 64, False), (1, 1, 32, True), (64, 64, 32, False)])
def test_gemm_nt_split(M, N, K, abf16):
    """sept_gemm_nt_split (bf16 MFMA, hi/lo-split operands) vs a float64 product: the split keeps
    ~2^-17 relative per product, i.e. fp32-GEMM quality (the GRU layer-0 projections)."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    if abf16:
        x = x.bfloat16()
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    want = x.double() @ W.double().t() + b.double()
    y = ops.linear_nt_split(x.cuda(), W.cuda(), b.cuda()).cpu()
    assert (y.double() - want).abs().max() < 2e-5 * (1 + want.abs().max())
    yb = ops.linear_nt_split(x.cuda(), W.cuda(), None, out_dtype=torch.bfloat16).cpu()
    assert torch.allclose(yb.float(), (want - b.double()).float(), rtol=1e-2, atol=1e-2)
    # rows given through a view with a larger leading dimension
    xw = torch.zeros(M, K + 8, dtype=x.dtype)
    xw[:, :K] = x
    y2 = ops.linear_nt_split(xw.cuda()[:, :K], W.cuda(), b.cuda()).cpu()
    assert torch.equal(y2, y)
    with pytest.raises(Exception):
        ops.linear_nt_split(x.cuda()[:, 8:], W.cuda()[:, 8:])          # K not a multiple of 32


@pytest.mark.parametrize("S,N,K,xbf16", [(5600, 384, 1280, True), (5600, 192, 64, False), (5600, 384, 128, False),
                                         (777, 36, 40, True), (600, 4, 132, False), (512, 128, 8, True)])
def test_gemm_tn_split(S, N, K, xbf16):
    """sept_gemm_tn_split (dW = dy^T x on the bf16 MFMA with hi/lo-split operands, split-K) vs
    float64; also views with an offset / larger leading dimension (the dW_hh products) and
    run-to-run determinism."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(S + N + K)
    dy = torch.randn(S, N, generator=g)
    x = torch.randn(S, K, generator=g)
    if xbf16:
        x = x.bfloat16()
    want = dy.double().t() @ x.double()
    got = ops.linear_backward_weight(dy.cuda(), x.cuda())
    assert (got.cpu().double() - want).abs().max() < 3e-5 * S ** 0.5 * (1 + want.abs().max() / S ** 0.5)
    assert torch.equal(got, ops.linear_backward_weight(dy.cuda(), x.cuda()))
    if N % 8 == 0 and K % 16 == 0:
        dyw, xw = dy.cuda(), x.cuda()
        got2 = ops.linear_backward_weight(dyw[:, N // 2:], xw[:, K // 2:])
        assert torch.allclose(got2, got[N // 2:, K // 2:], rtol=1e-5, atol=1e-4)
    # the bias gradient (column sums of dy) from the same launch: same dW bit for bit, sums against float64, an output
    # slot that is not 16-byte aligned (a parameter's place in the flat gradient buffer), and a column-slice view of dy
    slot = torch.full((N + 3,), 7.0, device="cuda")
    got3 = ops.linear_backward_weight(dy.cuda(), x.cuda(), colsum_out=slot[1:N + 1])
    assert torch.equal(got3, got)
    want_cs = dy.double().sum(0)
    assert (slot[1:N + 1].cpu().double() - want_cs).abs().max() < 3e-6 * S ** 0.5 * (1 + want_cs.abs().max() / S ** 0.5)
    assert float(slot[0]) == 7.0 and float(slot[N + 1]) == 7.0           # nothing written outside the slot
    cs2 = torch.empty(N, device="cuda")
    ops.linear_backward_weight(dy.cuda(), x.cuda(), colsum_out=cs2)
    assert torch.equal(cs2, slot[1:N + 1])                               # deterministic
    if N % 8 == 0 and K % 16 == 0:
        cs3 = torch.empty(N - N // 2, device="cuda")
        ops.linear_backward_weight(dyw[:, N // 2:], xw[:, K // 2:], colsum_out=cs3)
        assert torch.allclose(cs3, cs2[N // 2:], rtol=1e-5, atol=1e-4)


def test_gru_pack_matches_permute_and_transpose():
    from sept_amd import functional as SF
    g = torch.Generator().manual_seed(5)
    C, Wd = 128, 10
    wf, wr = torch.randn(192, C * Wd, generator=g), torch.randn(192, C * Wd, generator=g)
    bf, br = torch.randn(192, generator=g), torch.randn(192, generator=g)
    for layer in (0, 1):
        wcat, bcat, wcatT = SF._gru_cat_weights(wf.cuda(), wr.cuda(), bf.cuda(), br.cuda(), layer, C, Wd)
        want = torch.cat([wf, wr])
        if layer == 0:   # reference feature order (c, w) -> NHWC (w, c)
            want = want.view(384, C, Wd).transpose(1, 2).reshape(384, C * Wd)
        assert torch.equal(wcat.cpu(), want) and torch.equal(wcatT.cpu(), want.t())
        assert torch.equal(bcat.cpu(), torch.cat([bf, br]))


@pytest.mark.parametrize("B,T,H", [(7, 25, 64), (4, 3, 64), (1, 1, 64), (5, 25, 128), (2, 2, 128)])
def test_gru_layer_forward_backward(B, T, H):
    """One bidirectional GRU layer (input projections by sept_gemm + recurrent kernel) vs nn.GRU, for the
    trainer's hidden size (64) and the class default (128)."""
    from sept_amd import ops
    torch.manual_seed(B)
    K, G = 48, 3 * H
    ref = nn.GRU(K, H, num_layers=1, batch_first=True, bidirectional=True)
    x = torch.randn(B, T, K, requires_grad=True)
    want, _ = ref(x)
    dout = torch.randn(B, T, 2 * H)
    want.backward(dout)
    P = {n: p.detach().cuda() for n, p in ref.named_parameters()}
    xc = x.detach().cuda().view(B * T, K)
    gi = torch.empty(B * T, 2 * G, device="cuda")
    ops.gemm_raw(xc, K, 1, P["weight_ih_l0"], 1, K, gi, 2 * G, B * T, G, K, P["bias_ih_l0"])
    ops.gemm_raw(xc, K, 1, P["weight_ih_l0_reverse"], 1, K, gi[:, G:], 2 * G, B * T, G, K, P["bias_ih_l0_reverse"])
    out, gates = ops.gru_forward(gi.view(B, T, 2, G), P["weight_hh_l0"], P["weight_hh_l0_reverse"],
                                 P["bias_hh_l0"], P["bias_hh_l0_reverse"])
    assert torch.allclose(out.cpu(), want.detach(), rtol=1e-4, atol=1e-5)
    dgi, dgh, hprev = ops.gru_backward(dout.cuda(), out, gates, P["weight_hh_l0"], P["weight_hh_l0_reverse"])
    dgi2, dgh2, hp2 = dgi.view(B * T, 2 * G), dgh.view(B * T, 2 * G), hprev.view(B * T, 2 * H)
    for d, tag in ((0, ""), (1, "_reverse")):
        gs, gh = dgi2[:, d * G:(d + 1) * G], dgh2[:, d * G:(d + 1) * G]
        grads = dict(ref.named_parameters())
        chk = [("weight_ih_l0" + tag, ops.linear_backward_weight(gs, xc)),
               ("weight_hh_l0" + tag, ops.linear_backward_weight(gh, hp2[:, d * H:(d + 1) * H])),
               ("bias_ih_l0" + tag, ops.colsum(gs)), ("bias_hh_l0" + tag, ops.colsum(gh))]
        for name, got in chk:
            w = grads[name].grad
            assert torch.allclose(got.cpu(), w, rtol=1e-3, atol=1e-5 + 1e-4 * w.abs().max()), name
    dx = torch.empty(B * T, K, device="cuda")
    ops.gemm_raw(dgi2, 2 * G, 1, P["weight_ih_l0"], K, 1, dx, K, B * T, K, G)
    ops.gemm_raw(dgi2[:, G:], 2 * G, 1, P["weight_ih_l0_reverse"], K, 1, dx, K, B * T, K, G, beta=1.0)
    assert torch.allclose(dx.cpu().view(B, T, K), x.grad, rtol=1e-3, atol=1e-5)


def test_cloak_and_loss_and_small_ops():
    from sept_amd import ops
    g = torch.Generator().manual_seed(3)
    B, Wn, Fm = 5, 20, 16
    x = torch.randn(B, Wn * Fm, generator=g)
    locs = (0.1 * torch.randn(1, Wn, Fm, generator=g)).requires_grad_()
    rhos = (-2 + 0.5 * torch.randn(1, Wn, Fm, generator=g)).requires_grad_()
    eps = 0.1 * torch.randn(1, Wn, Fm, generator=g)
    mask = (torch.rand(1, Wn, Fm, generator=g) > 0.3).float()
    for m in (None, mask):
        locs.grad = rhos.grad = None
        scales = (1 + torch.tanh(rhos)) / 2 * (10 - 0.01) + 0.01
        e = eps if m is None else eps * m
        want = (x.view(B, 1, Wn, Fm) if m is None else x.view(B, 1, Wn, Fm) * m) + locs + scales * e
        dxa, dxb = torch.randn(B, Wn * Fm, generator=g), torch.randn(B, Wn * Fm, generator=g)
        (want.view(B, -1) * (dxa - 0.1 * dxb)).sum().backward(retain_graph=True)
        (-0.05 * torch.log(scales.mean())).backward()
        mc = None if m is None else m.cuda()
        xn = ops.cloak_forward(x.cuda(), locs.detach().cuda(), rhos.detach().cuda(), eps.cuda(), mc, 0.01, 10.0)
        assert torch.allclose(xn.cpu(), want.detach().view(B, -1), rtol=1e-5, atol=1e-6)
        _, mean = ops.cloak_scales(rhos.detach().cuda(), 0.01, 10.0, want_scales=False, want_mean=True)
        assert float(mean) == pytest.approx(float(scales.mean()), rel=1e-5)
        dl, dr = ops.cloak_backward(dxa.cuda(), dxb.cuda(), -0.1, rhos.detach().cuda(), eps.cuda(), mc, 0.01, 10.0,
                                    scale_lambda=0.05, scale_mean=mean)
        assert torch.allclose(dl.cpu(), locs.grad, rtol=1e-4, atol=1e-6)
        assert torch.allclose(dr.cpu(), rhos.grad, rtol=1e-4, atol=1e-7)
    # weighted CE
    logits = torch.randn(9, 4, generator=g, requires_grad=True)
    lab = torch.randint(0, 4, (9,), generator=g)
    w = 1 + torch.rand(9, generator=g)
    want = (F.cross_entropy(logits, lab, reduction="none") * w).sum() * (0.1 / 9)
    want.backward()
    loss = torch.zeros((), device="cuda")
    d = ops.cross_entropy(logits.detach().cuda(), lab.cuda(), w.cuda(), 0.1 / 9, loss)
    assert float(loss) == pytest.approx(float(want), rel=1e-5)
    assert torch.allclose(d.cpu(), logits.grad, rtol=1e-4, atol=1e-7)
    # mean over time, relu+dropout, permutation round trip, window/norm
    x3 = torch.randn(4, 25, 128, generator=g)
    assert torch.allclose(ops.mean_t_forward(x3.cuda()).cpu(), x3.mean(1), atol=1e-6)
    dz = torch.randn(4, 128, generator=g)
    assert torch.allclose(ops.mean_t_backward(dz.cuda(), 25).cpu(), (dz / 25)[:, None].expand(4, 25, 128), atol=1e-7)
    m2 = (torch.rand(4, 128, generator=g) > 0.2).float() / 0.8
    y = ops.relu_dropout_forward(dz.cuda(), m2.cuda()).cpu()
    assert torch.allclose(y, F.relu(dz) * m2)
    assert torch.allclose(ops.relu_dropout_backward(dz.cuda(), dz.cuda(), m2.cuda()).cpu(), (dz > 0) * dz * m2)
    Wm = torch.randn(6, 5 * 7, generator=g)
    pw = ops.permute_cols(Wm.cuda(), 5, 7)
    assert torch.equal(pw.cpu().view(6, 7, 5), Wm.view(6, 5, 7).transpose(1, 2))
    assert torch.equal(ops.permute_cols(pw, 5, 7, inverse=True).cpu(), Wm)
    mel = torch.randn(3, 501, 8, generator=g)
    mu, sd = torch.randn(8, generator=g), torch.rand(8, generator=g) + 0.5
    wn = ops.window_norm(mel.cuda(), mu.cuda(), sd.cuda()).cpu()
    assert wn.shape == (21, 200, 8)
    want = torch.stack([(mel[b, 50 * i:50 * i + 200] - mu) / (sd + 1e-5) for b in range(3) for i in range(7)])
    assert torch.allclose(wn, want, rtol=1e-5, atol=1e-6)
    short = ops.window_norm(mel[:, :120].cuda()).cpu()       # T < win: one zero-padded window
    assert short.shape == (3, 200, 8) and torch.equal(short[:, :120], mel[:, :120]) and (short[:, 120:] == 0).all()


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_optimizer_steps_match_torch(kind):
    from sept_amd import ops
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(1000, generator=g)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.SGD([ref], lr=1e-3, momentum=0.9, weight_decay=1e-4) if kind == "sgd" else \
        torch.optim.Adam([ref], lr=5e-4, weight_decay=1e-4, betas=(0.9, 0.98), eps=1e-9)
    p = p0.clone().cuda()
    b1, b2 = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        ref.grad = gr.clone()
        opt.step()
        if kind == "sgd":
            ops.sgd_step(p, (2 * gr).cuda(), b1, 1e-3, 0.9, 1e-4, step == 1, grad_scale=0.5)
        else:
            ops.adam_step(p, (2 * gr).cuda(), b1, b2, 5e-4, 0.9, 0.98, 1e-9, 1e-4, step, grad_scale=0.5)
        assert torch.allclose(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-7), step


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_device_state_optimizer_steps_match_torch_with_a_schedule(kind):
    """sept_sgd_step_dev / sept_adam_step_dev: learning rate and Adam's step count live on the device (so the
    update can sit inside a captured graph); driven by a StepLR schedule they follow torch.optim + StepLR."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(2)
    p0 = torch.randn(1000, generator=g)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.SGD([ref], lr=1e-3, momentum=0.9, weight_decay=1e-4) if kind == "sgd" else \
        torch.optim.Adam([ref], lr=5e-4, weight_decay=1e-4, betas=(0.9, 0.98), eps=1e-9)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)
    p = p0.clone().cuda()
    b1, b2 = torch.zeros_like(p), torch.zeros_like(p)
    lr_dev = torch.empty((), device="cuda")
    step_dev = torch.zeros((), dtype=torch.int64, device="cuda")
    for step in range(1, 7):
        gr = torch.randn(1000, generator=g)
        ref.grad = gr.clone()
        ops.fill(lr_dev, opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        ops.counter_add(step_dev, 1)
        if kind == "sgd":
            ops.sgd_step_dev(p, (2 * gr).cuda(), b1, lr_dev, 0.9, 1e-4, grad_scale=0.5)
        else:
            ops.adam_step_dev(p, (2 * gr).cuda(), b1, b2, lr_dev, 0.9, 0.98, 1e-9, 1e-4, step_dev, grad_scale=0.5)
        assert torch.allclose(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-7), step
    assert int(step_dev) == 6 and float(lr_dev) == pytest.approx(opt.param_groups[0]["lr"] * 2)   # 3 halvings due, 2 applied


def test_philox_dropout_and_normal_statistics():
    """sept_dropout_mask / sept_normal: right distribution, reproducible for equal (seed, counter),
    fresh after begin_step(), different sub-streams per call site."""
    from sept_amd import ops
    r = ops.Rng(1234, "cuda")
    r.begin_step()
    m = r.dropout_mask((1000, 1000), 0.2)
    vals = torch.unique(m)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and float(vals[1]) == pytest.approx(1.25)
    assert abs(float((m == 0).float().mean()) - 0.2) < 2e-3
    e = r.normal((1, 200, 128), 0.0, 0.1)
    assert abs(float(e.mean())) < 2e-3 and abs(float(e.std()) - 0.1) < 2e-3
    z = r.normal((1000, 1000))
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3 and float(z.abs().max()) < 7.0
    assert abs(float((z.abs() < 1.0).float().mean()) - 0.6827) < 3e-3
    r2 = ops.Rng(1234, "cuda")          # same seed, same counter, same call order -> identical draws
    r2.begin_step()
    assert torch.equal(r2.dropout_mask((1000, 1000), 0.2), m)
    assert torch.equal(r2.normal((1, 200, 128), 0.0, 0.1), e)
    r2.begin_step()                      # next step: fresh numbers
    assert not torch.equal(r2.dropout_mask((1000, 1000), 0.2), m)
    assert not torch.equal(ops.Rng(99, "cuda").dropout_mask((1000, 1000), 0.2), m)


def test_preprocess_speaker_stats_window_norm_and_augmentation():
    """Device side of preprocess_adversary_data.py:356-423 against numpy: per-speaker nanmean / nanstd /
    nanmin / nanmax, z-norm and min-max windows (also a short, zero-padded clip) and the Gaussian
    class-balance augmentation."""
    import numpy as np
    from sept_amd import preprocess as pp
    g = torch.Generator().manual_seed(11)
    B, T, F, S = 6, 301, 40, 3
    mel = torch.randn(B, T, F, generator=g) * 9 - 30
    spk = torch.tensor([0, 2, 0, 1, 2, 2], dtype=torch.int32)
    stats = pp.speaker_stats(mel.cuda(), spk.cuda(), S, population="frames").cpu().numpy()
    m = mel.double().numpy()
    for s_ in range(S):
        rows = m[(spk == s_).numpy()].reshape(-1, F)
        np.testing.assert_allclose(stats[s_, 0], rows.mean(0), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(stats[s_, 1], rows.std(0), rtol=1e-5)
        np.testing.assert_allclose(stats[s_, 2], rows.min(0), rtol=1e-6)
        np.testing.assert_allclose(stats[s_, 3], rows.max(0), rtol=1e-6)
    st = torch.from_numpy(stats).cuda()
    for norm in ("znorm", "min_max"):
        w = pp.window_normalize(mel.cuda(), st, spk.cuda(), norm).cpu().numpy()
        assert w.shape == (B * 3, 200, F)                      # int((301 - 200) / 50) + 1 = 3 windows
        for b in (0, 3, 5):
            for i in range(3):
                x = m[b, 50 * i:50 * i + 200]
                s_ = int(spk[b])
                want = (x - stats[s_, 0]) / (stats[s_, 1] + 1e-5) if norm == "znorm" else \
                    (x - stats[s_, 2]) / (stats[s_, 3] - stats[s_, 2]) * 2 - 1
                np.testing.assert_allclose(w[b * 3 + i], want, rtol=2e-4, atol=2e-4)
    short = pp.window_normalize(mel[:1, :120].contiguous().cuda(), st[:1], None, "znorm").cpu().numpy()
    assert short.shape == (1, 200, F)
    np.testing.assert_allclose(short[0, 120:], np.broadcast_to((0 - stats[0, 0]) / (stats[0, 1] + 1e-5), (80, F)),
                               rtol=1e-4, atol=1e-4)             # padded with zeros BEFORE the normalisation
    # augmentation: noise statistics and class balance
    x = torch.zeros(64, 200, F).cuda()
    noise = pp.add_gaussian(x, 0.05)
    assert abs(float(noise.mean())) < 1e-3 and float(noise.std()) == pytest.approx(0.05, rel=2e-2)
    labels = torch.tensor([0] * 40 + [1] * 14 + [2] * 10)
    xa, la = pp.balance_by_augmentation(x, labels, generator=torch.Generator().manual_seed(1))
    assert xa.shape[0] == 120 and torch.bincount(la.cpu()).tolist() == [40, 40, 40]
    assert float(xa[:64].abs().max()) == 0.0 and float(xa[64:].std()) == pytest.approx(0.05, rel=5e-2)


def test_speaker_stats_follow_the_reference_population():
    """preprocess_adversary_data.py:20-83, 356-381 restated in oracle/preprocess_oracle.py: the statistics run over
    the rows of the SAVED windows (multiplicity 0-4 per frame), test-split clips whole and once, short clips once;
    ragged clips (zero-padded to one T with `lengths`), then znorm / min-max of the stored items (zeros for the
    padding BEFORE the normalisation)."""
    import numpy as np
    from oracle import preprocess_oracle as po
    from sept_amd import preprocess as pp
    rng = np.random.default_rng(5)
    F, T = 24, 420
    lens = [420, 301, 200, 249, 120, 333, 199, 250]
    spk = [0, 1, 0, 2, 1, 2, 0, 1]
    test_speakers = {2}
    clips = [rng.normal(-30, 9, size=(L, F)) * (1 + 0.2 * s_) for L, s_ in zip(lens, spk)]
    want = po.speaker_statistics(clips, spk, test_speakers)
    mel = torch.zeros(len(lens), T, F)
    for b, c in enumerate(clips):
        mel[b, :len(c)] = torch.from_numpy(c).float()
    whole = torch.tensor([s_ in test_speakers for s_ in spk])
    stats = pp.speaker_stats(mel.cuda(), torch.tensor(spk), 3, lengths=torch.tensor(lens), whole_clip=whole).cpu().numpy()
    for s_ in range(3):
        np.testing.assert_allclose(stats[s_, 0], want[s_]["mean"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(stats[s_, 1], want[s_]["std"], rtol=1e-5)
        np.testing.assert_allclose(stats[s_, 2], want[s_]["min"], rtol=1e-6)
        np.testing.assert_allclose(stats[s_, 3], want[s_]["max"], rtol=1e-6)
    # the population really differs from "every frame once" (what round 1 computed)
    frames = pp.speaker_stats(mel[:1].cuda(), None, 1, population="frames").cpu().numpy()
    windows = pp.speaker_stats(mel[:1].cuda(), None, 1).cpu().numpy()
    assert np.abs(frames[0, 0] - windows[0, 0]).max() > 1e-3
    mult = po.frame_multiplicity(420)
    assert mult.max() == 4 and mult[-20:].sum() == 0 and mult.sum() == 5 * 200
    # windows of a full-length clip and of a short one, normalised with the reference-population statistics
    st = torch.from_numpy(stats).cuda()
    for norm in ("znorm", "min_max"):
        for b in (0, 4):
            L = lens[b]
            w = pp.window_normalize(mel[b:b + 1, :L].contiguous().cuda(), st, torch.tensor([spk[b]]), norm).cpu().numpy()
            items = po.saved_items(clips[b])
            assert w.shape[0] == len(items)
            for i, (_, stored) in enumerate(items):
                np.testing.assert_allclose(w[i], po.normalise(stored, want[spk[b]], norm), rtol=3e-4, atol=3e-4)


def test_preprocess_kernels_reproduce_the_reference_run(golden_dir):
    """sept_speaker_stats_windows + sept_window_norm_spk against what the REFERENCE's own write_data_dict /
    save_data_dict / normalisation block produced on the synthetic clips of tests/preprocess_synth.py
    (tests/golden/preprocess_golden.npz, tools/make_goldens_preprocess.py): statistics of every speaker and every stored
    item of the train / validation / adversary splits (windows; short clips zero padded) in both normalisation modes.
    Test-split clips are stored whole by the reference (no windows): their frames enter the statistics once each."""
    import os
    import numpy as np
    from sept_amd import preprocess as pp
    from tests.preprocess_synth import CLIPS, F, WIN, synthetic_clips
    G = np.load(os.path.join(golden_dir, "preprocess_golden.npz"))
    clips = synthetic_clips()
    names = [str(s) for s in G["stat_speakers"]]
    T = max(c[0] for c in CLIPS)
    mel = torch.zeros(len(CLIPS), T, F)
    for b, c in enumerate(clips):
        mel[b, :len(c)] = torch.from_numpy(c).float()
    spk = torch.tensor([names.index(c[1]) for c in CLIPS])
    lens = torch.tensor([c[0] for c in CLIPS])
    whole = torch.tensor([c[2] == "test" for c in CLIPS])
    stats = pp.speaker_stats(mel.cuda(), spk, len(names), lengths=lens, whole_clip=whole)
    got = stats.cpu().numpy()
    np.testing.assert_allclose(got[:, 0], G["stats"][:, 0], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(got[:, 1], G["stats"][:, 1], rtol=1e-5)
    np.testing.assert_allclose(got[:, 2], G["stats"][:, 2], rtol=1e-6)
    np.testing.assert_allclose(got[:, 3], G["stats"][:, 3], rtol=1e-6)
    st = torch.from_numpy(G["stats"].astype(np.float32)).cuda()     # the reference's own statistics from here on
    for norm in ("znorm", "min_max"):
        ref = {str(k): (int(n), c, e) for k, n, c, e in zip(G[f"{norm}_keys"], G[f"{norm}_len"], G[f"{norm}_sums"],
                                                            G[f"{norm}_edges"])}
        checked = 0
        for b, (L, name, split) in enumerate(CLIPS):
            if split == "test":
                continue
            w = pp.window_normalize(mel[b:b + 1, :L].contiguous().cuda(), st, spk[b:b + 1], norm).double().cpu().numpy()
            n_items = sum(1 for k in ref if k.startswith(f"clip{b:02d}_"))
            assert w.shape == (n_items, WIN, F), (b, w.shape, n_items)
            for i in range(n_items):
                n, sums, edges = ref[f"clip{b:02d}_{i}"]
                assert n == WIN
                scale = max(1.0, float(np.abs(w[i]).max()))
                np.testing.assert_allclose(np.concatenate([w[i, :4], w[i, -4:]]), edges, rtol=2e-4, atol=2e-4 * scale)
                assert abs(w[i].sum() - sums[0]) < 1e-4 * sums[1] + 1e-3 and abs(np.abs(w[i]).sum() - sums[1]) < 1e-4 * sums[1]
                if f"{norm}_clip{b:02d}_{i}" in G.files:
                    np.testing.assert_allclose(w[i], G[f"{norm}_clip{b:02d}_{i}"][0], rtol=2e-4, atol=2e-4 * scale)
                checked += 1
        assert checked == 27


@pytest.mark.parametrize("B,T,H", [(7, 25, 64), (3, 2, 64), (1, 1, 64), (5, 25, 128), (2, 3, 128)])
def test_lstm_layer_forward_backward(B, T, H):
    """One bidirectional LSTM layer (input projections by sept_gemm + sept_lstm_forward / backward) vs nn.LSTM
    (the default cell of deep_two_d_cnn_lstm_tmp, baseline_models.py:388-509)."""
    from sept_amd import ops
    torch.manual_seed(B + H)
    K, G = 48, 4 * H
    ref = nn.LSTM(K, H, num_layers=1, batch_first=True, bidirectional=True)
    x = torch.randn(B, T, K, requires_grad=True)
    want, _ = ref(x)
    dout = torch.randn(B, T, 2 * H)
    want.backward(dout)
    P = {n: p.detach().cuda() for n, p in ref.named_parameters()}
    xc = x.detach().cuda().view(B * T, K)
    gi = torch.empty(B * T, 2 * G, device="cuda")
    ops.gemm_raw(xc, K, 1, P["weight_ih_l0"], 1, K, gi, 2 * G, B * T, G, K, P["bias_ih_l0"])
    ops.gemm_raw(xc, K, 1, P["weight_ih_l0_reverse"], 1, K, gi[:, G:], 2 * G, B * T, G, K, P["bias_ih_l0_reverse"])
    out, gates, cells = ops.lstm_forward(gi.view(B, T, 2, G), P["weight_hh_l0"], P["weight_hh_l0_reverse"],
                                         P["bias_hh_l0"], P["bias_hh_l0_reverse"])
    assert torch.allclose(out.cpu(), want.detach(), rtol=1e-4, atol=1e-5)
    dg, hprev = ops.lstm_backward(dout.cuda(), out, gates, cells, P["weight_hh_l0"], P["weight_hh_l0_reverse"])
    dg2, hp2 = dg.view(B * T, 2 * G), hprev.view(B * T, 2 * H)
    grads = dict(ref.named_parameters())
    for d, tag in ((0, ""), (1, "_reverse")):
        gs = dg2[:, d * G:(d + 1) * G]
        chk = [("weight_ih_l0" + tag, ops.linear_backward_weight(gs, xc)),
               ("weight_hh_l0" + tag, ops.linear_backward_weight(gs, hp2[:, d * H:(d + 1) * H])),
               ("bias_ih_l0" + tag, ops.colsum(gs)), ("bias_hh_l0" + tag, ops.colsum(gs))]
        for name, got in chk:
            w = grads[name].grad
            assert torch.allclose(got.cpu(), w, rtol=1e-3, atol=1e-5 + 1e-4 * w.abs().max()), name
    dx = torch.empty(B * T, K, device="cuda")
    ops.gemm_raw(dg2, 2 * G, 1, P["weight_ih_l0"], K, 1, dx, K, B * T, K, G)
    ops.gemm_raw(dg2[:, G:], 2 * G, 1, P["weight_ih_l0_reverse"], K, 1, dx, K, B * T, K, G, beta=1.0)
    assert torch.allclose(dx.cpu().view(B, T, K), x.grad, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("T, masked, normed", [(501, False, True), (501, True, True), (120, True, False)])
def test_windows_formed_inside_the_cloak_kernel(T, masked, normed):
    """sept_window_norm_cloak (the fused pipeline's step input: preprocess_adversary_data.py:30-35,131 windows +
    cloak_models.py:45-58 noise in one pass) is bit-identical to sept_window_norm followed by sept_cloak_forward,
    short (zero-padded) clips included."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(T)
    B, F, win, shift = 3, 80, 200, 50
    mel = torch.randn(B, T, F, generator=g).cuda()
    mean = torch.randn(F, generator=g).cuda() if normed else None
    std = (torch.rand(F, generator=g) + 0.5).cuda() if normed else None
    locs = (0.1 * torch.randn(1, win, F, generator=g)).cuda()
    rhos = torch.randn(1, win, F, generator=g).cuda()
    eps = (0.1 * torch.randn(1, win, F, generator=g)).cuda()
    mask = (torch.rand(1, win, F, generator=g) > 0.3).float().cuda() if masked else None
    lw = ops.LazyWindows(mel, mean, std, win, shift)
    got = ops.window_norm_cloak(lw, locs, rhos, eps, mask, 0.01, 10.0)
    x = ops.window_norm(mel, mean, std, win, shift)
    want = ops.cloak_forward(x.view(x.shape[0], -1), locs, rhos, eps, mask, 0.01, 10.0)
    assert lw.shape == (x.shape[0], 1, win, F) and torch.equal(lw.materialise().view_as(x), x)
    assert torch.equal(got, want)


def test_fused_operand_prep_equals_the_separate_entries():
    """sept_prepare_operands: conv1's operand block, 5x5 conv operands in both orientations and the packed recurrent
    matrices (layer 0 with its column permutation) from ONE launch, bit-identical to sept_conv1_prep /
    sept_conv5x5_prep_weights / sept_gru_pack; and functional.prepare_operands fills exactly the cache entries the
    forward / backward passes look up (a second call finds nothing stale)."""
    from sept_amd import functional as SF, ops
    g = torch.Generator().manual_seed(13)
    w1, b1 = torch.randn(32, 1, 5, 5, generator=g).cuda(), torch.randn(32, generator=g).cuda()
    w2, w3 = torch.randn(64, 32, 5, 5, generator=g).cuda(), torch.randn(128, 64, 5, 5, generator=g).cuda()
    C, Wd, H = 128, 10, 64
    wf, wr = torch.randn(3 * H, C * Wd, generator=g).cuda(), torch.randn(3 * H, C * Wd, generator=g).cuda()
    bf, br = torch.randn(3 * H, generator=g).cuda(), torch.randn(3 * H, generator=g).cuda()
    wf1, wr1 = torch.randn(3 * H, 2 * H, generator=g).cuda(), torch.randn(3 * H, 2 * H, generator=g).cuda()
    items = [("conv1", w1, b1), ("conv5x5", w2, 0), ("conv5x5", w2, 1), ("conv5x5", w3, 0), ("conv5x5", w3, 1),
             ("gru", wf, wr, bf, br, C, Wd), ("gru", wf1, wr1, bf, br, 0, 0), ("conv1", w1, None)]
    got = ops.prepare_operands(items, w1.device)
    assert torch.equal(got[0], ops.conv1_prep(w1, b1)) and torch.equal(got[7], ops.conv1_prep(w1, None))
    for k, (w, mode) in zip((1, 2, 3, 4), ((w2, 0), (w2, 1), (w3, 0), (w3, 1))):
        assert torch.equal(got[k], ops.conv5x5_prep_weights(w, mode))
    for k, (a, b, layer) in zip((5, 6), ((wf, wr, 0), (wf1, wr1, 1))):
        wcat, bcat, wcatT = SF._gru_cat_weights(a, b, bf, br, layer, C, Wd)
        assert torch.equal(got[k][0], wcat) and torch.equal(got[k][1], bcat) and torch.equal(got[k][2], wcatT)
    # through the cache of a real network
    from model import baseline_models as bm
    m = bm.two_d_cnn_lstm(1, 80, 64, lstm_hidden_size=64, num_layers_lstm=2, pred="gender", attention_size=128, att=None,
                          global_feature=0).cuda()
    P = SF.trunk_params(m, "gender")
    assert SF.prepare_operands(P, 80) == 7            # conv1 + 2 x 2 conv operands + 2 recurrent packs
    assert SF.prepare_operands(P, 80) == 0            # nothing stale any more
    assert torch.equal(SF._conv1_operand(P.convs[0]), ops.conv1_prep(m.conv[0].weight, m.conv[0].bias))
    hit = SF._cached("convdgrad", m.conv[10].weight, lambda: None)
    assert hit is not None and torch.equal(hit, ops.conv5x5_prep_weights(m.conv[10].weight, 1))
    SF.invalidate_weight_cache()
    assert SF.prepare_operands(P, 80, need_dgrad=False) == 5


@pytest.mark.parametrize("NC,use_w,drop", [(4, True, True), (2, False, False), (6, True, False)])
def test_head_backward_with_the_cross_entropy_gradient_formed_in_the_kernel(NC, use_w, drop):
    """sept_head_backward_ce == sept_cross_entropy + sept_head_backward bit for bit, and both follow torch autograd through
    mean_t -> dense1 -> ReLU (* dropout scale) -> prediction layer -> weighted cross-entropy (baseline_models.py:231-258,
    training_cloak_with_grl.py:143-151)."""
    from sept_amd import ops
    torch.manual_seed(5)
    B, T, D, D1 = 37, 25, 128, 128
    x = torch.randn(B, T, D)
    W1, b1 = torch.randn(D1, D) * 0.1, torch.randn(D1) * 0.1
    Wh, bh = torch.randn(NC, D1) * 0.2, torch.randn(NC) * 0.1
    labels = torch.randint(0, NC, (B,))
    w = (torch.rand(B) + 0.5) if use_w else None
    mask = ((torch.rand(B, D1) > 0.2).float() / 0.8) if drop else None
    scale = 0.1 / B
    c = lambda t: None if t is None else t.cuda()
    logits, z, d1, d1a = ops.head_forward(c(x), c(W1), c(b1), c(mask), c(Wh), c(bh))
    loss = torch.zeros((), device="cuda")
    dl = ops.cross_entropy(logits, c(labels), c(w), scale, loss)
    dd1, dx = ops.head_backward(dl, c(Wh), d1, c(mask), c(W1), T)
    dl2, dd1_2, dx2 = ops.head_backward_ce(logits, c(labels), c(w), scale, c(Wh), d1, c(mask), c(W1), T)
    assert torch.equal(dl, dl2) and torch.equal(dd1, dd1_2) and torch.equal(dx, dx2)
    xr = x.clone().requires_grad_(True)
    a = torch.relu(xr.mean(1) @ W1.T + b1)
    if mask is not None:
        a = a * mask
    lg = a @ Wh.T + bh
    ce = torch.nn.functional.cross_entropy(lg, labels, reduction="none")
    ref = scale * ((ce * w).sum() if w is not None else ce.sum())
    ref.backward()
    torch.testing.assert_close(logits.cpu(), lg.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(loss.cpu(), ref.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(dx2.cpu(), xr.grad, rtol=1e-3, atol=1e-8)
