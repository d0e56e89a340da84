"""GPU parity of the conv1 (1->32) kernels and the fused BatchNorm+ReLU+MaxPool+Dropout2d
kernels against plain fp32 torch ops on the same (bf16-rounded) operands.
Reference layers: model/baseline_models.py:172-176."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("B,H,W", [(3, 200, 80), (2, 200, 128), (1, 37, 23)])
def test_conv1_forward_backward(B, H, W):
    from sept_amd import ops
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, H, W, generator=g).cuda()
    w = (torch.randn(32, 1, 5, 5, generator=g) * 0.2).cuda()
    bias = (torch.randn(32, generator=g) * 0.1).cuda()
    y = ops.conv1_forward(x, w, bias).float()
    want = nhwc(F.conv2d(x[:, None], w, bias, padding=2))
    assert torch.allclose(y, want, rtol=1e-2, atol=1e-2)
    assert ((y - want).abs() <= want.abs() * 2 ** -7 + 1e-4).all()   # fp32 math, one bf16 rounding
    # data gradient
    dy = torch.randn(B, H, W, 32, generator=g).bfloat16().cuda()
    dx = ops.conv1_backward_data(dy, w)
    xr = x.clone().requires_grad_()
    F.conv2d(xr[:, None], w.bfloat16().float(), None, padding=2).backward(nchw(dy.float()))
    assert torch.allclose(dx, xr.grad, rtol=1e-4, atol=1e-4), (dx - xr.grad).abs().max()
    # weight gradient: on MFMA when W % 8 == 0 (x rounded to bf16 like every MFMA operand), so the
    # fp32 reference sees the same rounded x
    dw, db = ops.conv1_backward_weight(x, dy)
    wr = w.clone().requires_grad_()
    br = bias.clone().requires_grad_()
    xq = x.bfloat16().float() if W % 8 == 0 else x
    F.conv2d(xq[:, None], wr, br, padding=2).backward(nchw(dy.float()))
    scale = wr.grad.abs().max()
    assert torch.allclose(dw, wr.grad, rtol=1e-3, atol=1e-4 * scale), (dw - wr.grad).abs().max()
    assert torch.allclose(db, br.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("B,H,W,C,pool", [(4, 200, 80, 32, 2), (3, 100, 40, 64, 2), (3, 50, 20, 128, 2),
                                          (2, 25, 10, 128, 1), (2, 51, 21, 64, 2)])
@pytest.mark.parametrize("drop", [False, True])
@pytest.mark.parametrize("pooled", [False, True])
def test_bn_relu_pool_forward_backward(B, H, W, C, pool, drop, pooled):
    """pooled: the backward takes its channel sums from the pooled output y (the training path) instead of
    from every window of x; channels with a tiny |gamma| must fall back to the windows."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, H, W, C, generator=g) * 1.5 + 0.3).bfloat16().cuda()
    gamma = (1 + 0.2 * torch.randn(C, generator=g))
    gamma[3], gamma[C - 5] = 1e-5, -2e-4
    gamma = gamma.cuda()
    beta = (0.2 * torch.randn(C, generator=g)).cuda()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    nbt = torch.zeros((), dtype=torch.int64).cuda()
    ds = None
    if drop:
        ds = ((torch.rand(B, C, generator=g) > 0.2).float() / 0.8).cuda()
    mean, invstd = ops.bn_stats(x, rm, rv, nbt)
    y16 = ops.bn_relu_pool_forward(x, mean, invstd, gamma, beta, ds, pool)
    y = y16.float()

    # reference on the CPU in fp32 (torch-ROCm's GPU batch_norm backward returns a dbeta that
    # disagrees with the sum of its own incoming gradient for odd widths -- observed on this
    # image -- so the GPU eager op is not used as the checker)
    xr = nchw(x.float().cpu()).requires_grad_()
    gr, br = gamma.cpu().clone().requires_grad_(), beta.cpu().clone().requires_grad_()
    rm2, rv2 = torch.zeros(C), torch.ones(C)
    t = F.relu(F.batch_norm(xr, rm2, rv2, gr, br, training=True, momentum=0.1, eps=1e-5))
    if pool == 2:
        t = F.max_pool2d(t, 2, 2)
    if drop:
        t = t * ds.cpu()[:, :, None, None]
    want = nhwc(t)
    assert torch.allclose(mean.cpu(), nchw(x.float().cpu()).mean((0, 2, 3)), atol=1e-5)
    assert torch.allclose(rm.cpu(), rm2, atol=1e-6) and torch.allclose(rv.cpu(), rv2, rtol=1e-5) and int(nbt) == 1
    assert torch.allclose(y.cpu(), want.detach(), rtol=1e-2, atol=1e-2)
    dy = torch.randn(want.shape, generator=g).bfloat16().cuda()
    dx, dgamma, dbeta = ops.bn_relu_pool_backward(dy, x, mean, invstd, gamma, beta, ds, pool,
                                                  y=y16 if pooled else None)
    t.backward(nchw(dy.float().cpu()))
    wdx = nhwc(xr.grad)
    # dx is rounded to bf16 once: per-element bound of one bf16 ulp + a small absolute term
    assert ((dx.float().cpu() - wdx).abs() <= wdx.abs() * 2 ** -7 + 1e-3 * wdx.abs().max()).all()
    # pooled sums: xhat at the maximum is recovered from the bf16 pooled output (one more bf16 rounding,
    # 2^-9 relative, random sign), so dgamma carries a few 1e-3 of noise -- the same size as the rounding
    # the bf16 pre-activations already have against an fp32 network; dbeta does not depend on xhat
    tol = 5e-3 if pooled else 1e-3
    assert torch.allclose(dgamma.cpu(), gr.grad, rtol=tol, atol=tol * gr.grad.abs().max())
    assert torch.allclose(dbeta.cpu(), br.grad, rtol=1e-3, atol=1e-3 * br.grad.abs().max())


def test_bn_eval_stats():
    from sept_amd import ops
    rm, rv = torch.randn(64).cuda(), (torch.rand(64) + 0.5).cuda()
    mean, invstd = ops.bn_eval_stats(rm, rv)
    assert torch.equal(mean, rm) and torch.allclose(invstd, (rv + 1e-5).rsqrt(), rtol=1e-6)


def test_sync_bn_split_entries_equal_global_batch():
    """Sync-BN entry points (SURVEY.md 8e option 1): two 'ranks' = two halves of a batch on one GPU.
    Summing their per-channel sums (what the all-reduce does) must reproduce the statistics, dx and
    parameter gradients of the fused single-process pass over the whole batch."""
    from sept_amd import ops
    from sept_amd._lib import lib, check
    g = torch.Generator().manual_seed(21)
    B, H, W, C, pool = 4, 20, 12, 64, 2
    x = (torch.randn(B, H, W, C, generator=g) * 1.5 + 0.3).bfloat16().cuda()
    dy = torch.randn(B, H // pool, W // pool, C, generator=g).bfloat16().cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    drop = (torch.rand(B, C, generator=g) > 0.2).float().cuda() * 1.25
    s = torch.cuda.current_stream().cuda_stream
    ws = ops.workspace("bn", lib.sept_bn_workspace_floats(C), x.device)
    mean_f, invstd_f = ops.bn_stats(x)
    halves = [slice(0, 2), slice(2, 4)]
    tot = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
    for h in halves:
        xs = x[h].contiguous()
        sums = torch.empty(2 * C, dtype=torch.float64, device="cuda")
        check(lib.sept_bn_partial_sums(xs.data_ptr(), xs.numel() // C, C, ws.data_ptr(), sums.data_ptr(), s), "sums")
        tot += sums
    mean, invstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    check(lib.sept_bn_stats_from_sums(tot.data_ptr(), float(B * H * W), C, mean.data_ptr(), invstd.data_ptr(), None, None,
                                      None, 0.1, 1e-5, s), "from_sums")
    assert torch.allclose(mean, mean_f, rtol=1e-6, atol=1e-6) and torch.allclose(invstd, invstd_f, rtol=1e-6)
    dx_f, dg_f, db_f = ops.bn_relu_pool_backward(dy, x, mean_f, invstd_f, gamma, beta, drop, pool)
    parts, tot_b = [], torch.zeros(2 * C, device="cuda")
    for h in halves:
        xs, dys, dr = x[h].contiguous(), dy[h].contiguous(), drop[h].contiguous()
        sums, dg, db = torch.empty(2 * C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        check(lib.sept_bn_relu_pool_backward_reduce(dys.data_ptr(), xs.data_ptr(), None, mean_f.data_ptr(), invstd_f.data_ptr(),
                                                    gamma.data_ptr(), beta.data_ptr(), dr.data_ptr(), ws.data_ptr(),
                                                    sums.data_ptr(), dg.data_ptr(), db.data_ptr(), 2, H, W, C, pool, s), "red")
        parts.append((xs, dys, dr, dg, db))
        tot_b += sums
    dxs = []
    for xs, dys, dr, dg, db in parts:
        dx = torch.empty_like(xs)
        check(lib.sept_bn_relu_pool_backward_apply(dys.data_ptr(), xs.data_ptr(), mean_f.data_ptr(), invstd_f.data_ptr(),
                                                   gamma.data_ptr(), beta.data_ptr(), dr.data_ptr(), tot_b.data_ptr(),
                                                   float(B * H * W), dx.data_ptr(), 2, H, W, C, pool, s), "apply")
        dxs.append(dx)
    assert torch.allclose(torch.cat(dxs).float(), dx_f.float(), rtol=2e-2, atol=1e-3)
    assert (torch.cat(dxs).float() - dx_f.float()).abs().mean() < 1e-5
    assert torch.allclose(parts[0][3] + parts[1][3], dg_f, rtol=1e-4, atol=1e-4)
    assert torch.allclose(parts[0][4] + parts[1][4], db_f, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,H,W", [(3, 200, 80), (2, 37, 23), (1, 16, 128)])
def test_conv1_forward_with_fused_bn_statistics(B, H, W):
    """sept_conv1_forward_stats: same output as the plain forward, and the statistics it leaves equal a
    BatchNorm statistics pass over that output (mean / invstd and the running buffers)."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(W)
    x = torch.randn(B, H, W, generator=g).cuda()
    w = (torch.randn(32, 1, 5, 5, generator=g) * 0.2).cuda()
    bias = (torch.randn(32, generator=g) * 0.1).cuda()
    rm, rv, nb = torch.zeros(32).cuda(), torch.ones(32).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    y, mean, invstd = ops.conv1_forward_stats(x, w, bias, rm, rv, nb)
    assert torch.equal(y, ops.conv1_forward(x, w, bias))
    rm2, rv2, nb2 = torch.zeros(32).cuda(), torch.ones(32).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    mean2, invstd2 = ops.bn_stats(y, rm2, rv2, nb2)
    assert torch.allclose(mean, mean2, rtol=1e-5, atol=1e-6) and torch.allclose(invstd, invstd2, rtol=1e-5)
    assert torch.allclose(rm, rm2, rtol=1e-5, atol=1e-7) and torch.allclose(rv, rv2, rtol=1e-5) and int(nb) == 1


@pytest.mark.parametrize("B,H,W,cin,cout", [(5, 100, 40, 32, 64), (3, 50, 20, 64, 128), (2, 25, 10, 128, 128),
                                            (3, 37, 13, 32, 64), (1, 9, 5, 64, 128)])
def test_conv5x5_forward_with_fused_bn_statistics(B, H, W, cin, cout):
    """sept_conv5x5_forward_stats: same output as the plain forward (bit-equal), and the statistics it leaves
    equal a BatchNorm statistics pass over that output (mean / invstd and the running buffers)."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(B, H, W, cin, generator=g).bfloat16().cuda()
    w = (torch.randn(cout, cin, 5, 5, generator=g) * 0.05).cuda()
    bias = (torch.randn(cout, generator=g) * 0.3).cuda()
    wt = ops.conv5x5_prep_weights(w, 0)
    y0 = ops.conv5x5(x, wt, bias)
    rm0, rv0, n0 = torch.zeros(cout).cuda(), torch.ones(cout).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    mean0, invstd0 = ops.bn_stats(y0, rm0, rv0, n0)
    rm1, rv1, n1 = torch.zeros(cout).cuda(), torch.ones(cout).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    res = ops.conv5x5_forward_stats(x, wt, bias, rm1, rv1, n1)
    assert res is not None
    y1, mean1, invstd1 = res
    assert torch.equal(y0, y1)
    assert torch.allclose(mean1, mean0, rtol=1e-5, atol=1e-6), (mean1 - mean0).abs().max()
    assert torch.allclose(invstd1, invstd0, rtol=1e-5), (invstd1 / invstd0 - 1).abs().max()
    assert torch.allclose(rm1, rm0, rtol=1e-5, atol=1e-7) and torch.allclose(rv1, rv0, rtol=1e-5) and int(n1) == 1
    # the data-gradient shapes (cin > cout) have no statistics form
    assert ops.conv5x5_forward_stats(torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16).cuda(),
                                     torch.zeros(25, 32, 64, dtype=torch.bfloat16).cuda(), None) is None


@pytest.mark.parametrize("B,H,W", [(3, 200, 80), (2, 200, 128), (2, 34, 16), (5, 16, 48)])
@pytest.mark.parametrize("drop", [False, True])
def test_block1_in_one_pass_with_given_statistics_equals_the_separate_kernels(B, H, W, drop):
    """sept_conv1_bn_relu_pool_forward (the inference form of block 1: conv1 -> BatchNorm with given statistics -> ReLU ->
    MaxPool 2x2 -> Dropout2d scale in registers, baseline_models.py:172-176 in eval mode) against the separate conv1 +
    BatchNorm kernels, which the tests above hold to torch -- including a channel with a tiny and one with a negative
    gamma -- and against plain torch."""
    import torch.nn.functional as Fn
    from sept_amd import ops
    assert ops.conv1_fused_supported(H, W)
    assert not ops.conv1_fused_supported(H + 1, W) and not ops.conv1_fused_supported(H, W + 8)
    g = torch.Generator().manual_seed(H * W)
    x = (torch.randn(B, H, W, generator=g) * 1.2 + 0.1).cuda()
    w = (torch.randn(32, 1, 5, 5, generator=g) * 0.2).cuda()
    bias = (torch.randn(32, generator=g) * 0.1).cuda()
    gamma = 1 + 0.3 * torch.randn(32, generator=g)
    gamma[3], gamma[20] = 1e-5, -0.7
    gamma, beta = gamma.cuda(), (0.2 * torch.randn(32, generator=g)).cuda()
    dmask = ((torch.rand(B, 32, generator=g) > 0.2).float() * 1.25).cuda() if drop else None
    rmean, rvar = (0.1 * torch.randn(32, generator=g)).cuda(), (0.5 + torch.rand(32, generator=g)).cuda()
    mean, invstd = ops.bn_eval_stats(rmean, rvar, 1e-5)
    pre = ops.conv1_forward(x, w, bias)
    y = ops.bn_relu_pool_forward(pre, mean, invstd, gamma, beta, dmask, 2)
    y2 = ops.conv1_bn_relu_pool_forward(x, w, bias, mean, invstd, gamma, beta, dmask)
    assert y2.shape == y.shape
    differ = (y2.float() - y.float()).abs() > 0
    # the two paths fold the BatchNorm affine slightly differently (fma placement): a result may round the other way
    assert differ.float().mean() < 1e-3, float(differ.float().mean())
    assert ((y2.float() - y.float()).abs() <= y.float().abs() * 2 ** -7 + 1e-6).all()
    ref = Fn.max_pool2d(Fn.relu(Fn.batch_norm(Fn.conv2d(x[:, None], w, bias, padding=2), rmean, rvar, gamma, beta, False, 0.0, 1e-5)), 2)
    if dmask is not None:
        ref = ref * dmask[:, :, None, None]
    ref = ref.permute(0, 2, 3, 1)
    assert float((y2.float() - ref).norm() / ref.norm()) < 6e-3      # bf16 pre-activation + bf16 output


@pytest.mark.parametrize("B, H, W, drop", [(3, 16, 24, False), (2, 20, 80, True), (4, 8, 16, True), (2, 200, 80, False),
                                         (2, 12, 128, True)])
def test_block1_data_gradient_from_pooled_gradient_and_argmax_positions(B, H, W, drop):
    """sept_conv1_backward_data_sparse: block 1's backward pass (Dropout2d, MaxPool, ReLU, training-mode BatchNorm, conv1:
    baseline_models.py:172-176) down to the gradient of the network input WITHOUT a pre-activation-sized tensor -- the
    sparse part (pooled gradient at the recorded arg-max positions) through the MFMA data-gradient kernel, the dense part
    (BatchNorm's mean terms) as a linear map of the one-channel input (9 x 9 filter + the exact border ring).  Against
    torch autograd through the same chain in fp32 (arg-max decisions on the stored bf16 pre-activations, as the HIP
    forward takes them) and against the separate HIP passes."""
    import torch.nn.functional as Fn
    from sept_amd import ops
    g = torch.Generator().manual_seed(7 * H + W + B)
    x = torch.randn(B, H, W, generator=g).cuda()
    w = (torch.randn(32, 1, 5, 5, generator=g) * 0.25).cuda()
    bias = (0.1 * torch.randn(32, generator=g)).cuda()
    gamma = (1 + 0.3 * torch.randn(32, generator=g)).cuda()
    beta = (0.2 * torch.randn(32, generator=g)).cuda()
    dmask = ((torch.rand(B, 32, generator=g) > 0.2).float() * 1.25).cuda() if drop else None
    pre = ops.conv1_forward(x, w, bias)
    mean, invstd = ops.bn_stats(pre)
    y, idx = ops.bn_relu_pool_forward(pre, mean, invstd, gamma, beta, dmask, 2, want_argmax=True)
    assert torch.equal(y, ops.bn_relu_pool_forward(pre, mean, invstd, gamma, beta, dmask, 2))
    assert idx.dtype == torch.uint8 and int(idx.max()) <= 4
    dy = torch.randn(B, H // 2, W // 2, 32, generator=g).bfloat16().cuda()
    dx, dg, db = ops.conv1_backward_data_sparse(x, pre, dy, idx, mean, invstd, gamma, beta, dmask, w, bias, y=y)
    # the separate HIP passes (held to torch by the tests above)
    dpre_t, want_dg, want_db = ops.bn_relu_pool_backward(dy, pre, mean, invstd, gamma, beta, dmask, 2, y=y)
    hip_dx = ops.conv1_backward_data(dpre_t, w)
    assert torch.allclose(dg, want_dg, rtol=1e-5, atol=1e-6) and torch.allclose(db, want_db, rtol=1e-5, atol=1e-6)
    # torch autograd in fp32
    pr = pre.float().permute(0, 3, 1, 2).contiguous().requires_grad_()
    out = Fn.max_pool2d(Fn.relu(Fn.batch_norm(pr, None, None, gamma, beta, training=True, eps=1e-5)), 2)
    if dmask is not None:
        out = out * dmask[:, :, None, None]
    out.backward(dy.float().permute(0, 3, 1, 2))
    ref_dx = torch.nn.grad.conv2d_input((B, 1, H, W), w, pr.grad, padding=2)[:, 0]
    scale = float(ref_dx.abs().max())
    err_new = float((dx - ref_dx).norm() / ref_dx.norm())
    err_old = float((hip_dx - ref_dx).norm() / ref_dx.norm())
    assert err_new < 6e-3, (err_new, err_old)
    assert torch.allclose(dx, ref_dx, rtol=2e-2, atol=4e-3 * scale), float((dx - ref_dx).abs().max()) / scale
    assert float((dx - hip_dx).norm() / hip_dx.norm()) < 8e-3


@pytest.mark.parametrize("B,H,W", [(3, 200, 80), (2, 200, 128), (5, 38, 16)])
@pytest.mark.parametrize("drop", [False, True])
def test_block1_pool_first_equals_the_stored_tensor_path(B, H, W, drop):
    """sept_conv1_forward_pool + sept_bn_relu_ext_forward (round 3: the 2x2 window resolved BEFORE the BatchNorm, maximum
    or minimum by the sign of gamma; baseline_models.py:172-176) against the stored-tensor path (conv1 with statistics,
    BatchNorm + ReLU + MaxPool + Dropout2d with arg-max): identical statistics, bit-identical pooled activation, identical
    position bytes, ext = the window extremum of the stored pre-activations.  gamma has negative, tiny and zero entries;
    then the backward sums from (dy, ext, idx) -- by the reduce kernel and by the data-gradient conv's epilogue -- against
    the window-path sums of the stored tensor, and the data gradient against the pre-activation form."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(3 * H + W + B)
    x = torch.randn(B, H, W, generator=g).cuda()
    w = (torch.randn(32, 1, 5, 5, generator=g) * 0.25).cuda()
    bias = (0.1 * torch.randn(32, generator=g)).cuda()
    gamma = 1 + 0.3 * torch.randn(32, generator=g)
    gamma[2], gamma[7], gamma[11], gamma[30] = -0.8, 1e-5, -2e-4, 0.0
    gamma = gamma.cuda()
    beta = (0.2 * torch.randn(32, generator=g)).cuda()
    dmask = ((torch.rand(B, 32, generator=g) > 0.2).float() * 1.25).cuda() if drop else None
    pre, mean, invstd = ops.conv1_forward_stats(x, w, bias)
    y_old, idx_old = ops.bn_relu_pool_forward(pre, mean, invstd, gamma, beta, dmask, 2, want_argmax=True)
    rm, rv, nbt = torch.zeros(32).cuda(), torch.ones(32).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    ext, idx, mean2, invstd2 = ops.conv1_forward_pool(x, w, bias, gamma, rm, rv, nbt)
    assert int(nbt) == 1 and float(rm.abs().max()) > 0
    assert torch.allclose(mean2, mean, rtol=1e-5, atol=1e-6) and torch.allclose(invstd2, invstd, rtol=1e-5)
    # the extremum of the stored pre-activations, window by window
    p4 = pre.float().view(B, H // 2, 2, W // 2, 2, 32).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4, 32)
    want_ext = torch.where(gamma >= 0, p4.max(3).values, p4.min(3).values)
    assert torch.equal(ext.float(), want_ext)
    # first position attaining it, in scan order
    hit = p4 == want_ext[:, :, :, None, :]
    order = torch.arange(4, device=hit.device).view(1, 1, 1, 4, 1)
    want_pos = torch.where(hit, order, torch.full_like(order, 4)).min(3).values.to(torch.uint8)
    assert torch.equal(idx, want_pos)
    y_new = ops.bn_relu_ext_forward(ext, None, mean, invstd, gamma, beta, dmask)   # same statistics as the old path: bit equality
    assert torch.equal(y_new, y_old)
    assert torch.equal(idx, want_pos)             # the training path leaves the bytes pure positions
    # with the bytes handed in, the activation pass re-marks 4 where the ReLU is inactive: the old path's convention
    idx_m = idx.clone()
    assert torch.equal(ops.bn_relu_ext_forward(ext, idx_m, mean, invstd, gamma, beta, dmask), y_old)
    live = gamma != 0     # (gamma == 0: every position ties in the activations; ATen's rule picks 0, here the extremum of v)
    assert torch.equal(idx_m[..., live] == 4, idx_old[..., live] == 4)
    both = (idx_m != 4) & (idx_old != 4)
    assert torch.equal(idx_m[both & live], idx_old[both & live])
    # ---- backward sums: window path of the stored tensor vs (dy, ext); the reduce pass MASKS dy (zero where inactive) ----
    dy = torch.randn(B, H // 2, W // 2, 32, generator=g).bfloat16().cuda()
    _, want_dg, want_db = ops.bn_relu_pool_backward(dy, pre, mean, invstd, gamma, beta, dmask, 2)      # y=None: every window
    dy_raw = dy.clone()
    sums, dg, db = ops.bn_backward_sums_ext(dy, ext, mean, invstd, gamma, beta, dmask)
    assert torch.allclose(db[live], want_db[live], rtol=1e-4, atol=1e-4) and torch.allclose(dg[live], want_dg[live], rtol=1e-4, atol=1e-4)
    assert torch.equal(sums[:32], db) and torch.equal(sums[32:], dg)
    off = idx_m == 4
    assert torch.equal(dy[~off], dy_raw[~off]) and float(dy[off].abs().max()) == 0.0
    # ---- data gradient: sparse + dense form from (dy, idx, x, sums) vs the stored-tensor passes ----
    if W <= 128 and W >= 8 and H >= 4:
        dx = ops.conv1_backward_data_from_sums(x, dy, idx, sums, mean, invstd, gamma, dmask, w, bias)
        dpre_t, _, _ = ops.bn_relu_pool_backward(dy, pre, mean, invstd, gamma, beta, dmask, 2)
        hip_dx = ops.conv1_backward_data(dpre_t, w)
        assert float((dx - hip_dx).norm() / hip_dx.norm()) < 8e-3
        # ---- weight gradient: sparse product + Gram-matrix dense part vs (a) the stored-tensor passes, (b) torch autograd
        # in fp32 through BatchNorm (batch statistics) / ReLU / MaxPool / Dropout2d on the stored bf16 pre-activations ----
        import torch.nn.functional as Fn
        dw, db = ops.conv1_backward_weight_from_sums(x, dy, idx, sums, mean, invstd, gamma, dmask, w, bias)
        dw_old, db_old = ops.conv1_backward_weight(x, dpre_t)
        pr = pre.float().permute(0, 3, 1, 2).contiguous().requires_grad_()
        out = Fn.max_pool2d(Fn.relu(Fn.batch_norm(pr, None, None, gamma, beta, training=True, eps=1e-5)), 2)
        if dmask is not None:
            out = out * dmask[:, :, None, None]
        out.backward(dy.float().permute(0, 3, 1, 2))
        ref_dw = torch.nn.grad.conv2d_weight(x[:, None], (32, 1, 5, 5), pr.grad, padding=2)
        e_new = float((dw - ref_dw).norm() / ref_dw.norm())
        e_old = float((dw_old - ref_dw).norm() / ref_dw.norm())
        assert e_new < 1e-2, (e_new, e_old)
        # ---- the input gradient summed over the batch (all the cloak's backward needs), formed without a per-sample pass ----
        ref_dxs = torch.nn.grad.conv2d_input((B, 1, H, W), w, pr.grad, padding=2)[:, 0].sum(0)
        dxs = ops.conv1_backward_data_sum(x, dy, idx, sums, mean, invstd, gamma, dmask, w, bias)
        assert dxs.shape == (1, H, W)
        e_sum = float((dxs[0] - ref_dxs).norm() / ref_dxs.norm())
        e_per = float((dx.sum(0) - ref_dxs).norm() / ref_dxs.norm())
        assert e_sum < 2e-3 and e_sum <= e_per + 1e-4, (e_sum, e_per)     # fp32 weights, no bf16 rounding of scd * g
        assert float((dw - dw_old).norm() / dw_old.norm()) < 1.5e-2
        # the bias gradient of a conv in front of a train-mode BatchNorm vanishes: rounding noise only
        assert float(db.abs().max()) < 2e-2 * float(ref_dw.abs().max()) * 25


@pytest.mark.parametrize("B,H,W,cin,cout", [(3, 100, 40, 64, 32), (5, 100, 64, 64, 32)])
def test_dgrad_epilogue_sums_for_a_pool_first_block(B, H, W, cin, cout):
    """sept_conv5x5_dgrad_bnsums_ext: the data-gradient conv whose epilogue leaves (sum g, sum g * xhat) of the pool-first
    block in front of its output, from its output tile and that block's ext, and stores its output MASKED (zero where that
    block's ReLU is inactive): same as the plain conv followed by the masking reduce kernel."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(cin + H)
    dyo = torch.randn(B, H, W, cin, generator=g).bfloat16().cuda()
    wt = ops.conv5x5_prep_weights((torch.randn(cin, cout, 5, 5, generator=g) * 0.05).cuda(), 1)
    ext = (torch.randn(B, H, W, cout, generator=g) * 1.5).bfloat16().cuda()
    mean, invstd = (0.2 * torch.randn(cout, generator=g)).cuda(), (0.5 + torch.rand(cout, generator=g)).cuda()
    gamma, beta = (1 + 0.3 * torch.randn(cout, generator=g)).cuda(), (0.2 * torch.randn(cout, generator=g)).cuda()
    gamma[1] = -0.7
    dmask = ((torch.rand(B, cout, generator=g) > 0.2).float() * 1.25).cuda()
    dx, presums = ops.conv5x5_dgrad_bnsums_ext(dyo, wt, ext, mean, invstd, gamma, beta, dmask)
    assert presums is not None
    sums, dg, db = ops.bn_backward_sums_ext(dx, ext, mean, invstd, gamma, beta, dmask, presums)
    plain = ops.conv5x5(dyo, wt)
    sums2, dg2, db2 = ops.bn_backward_sums_ext(plain, ext, mean, invstd, gamma, beta, dmask, None)    # masks `plain` in place
    assert torch.equal(dx, plain)
    frac_off = float((plain == 0).float().mean())
    assert 0.2 < frac_off < 0.8                      # the mask really bites
    scale = float(sums2.abs().max())
    assert torch.allclose(sums, sums2, rtol=1e-4, atol=1e-5 * scale)
    assert torch.equal(dg, sums[cout:]) and torch.equal(db, sums[:cout])


@pytest.mark.parametrize("B,H,W,cin,cout,ep,drop", [
    (3, 100, 40, 64, 32, "ext", True), (2, 100, 40, 64, 32, None, False), (5, 50, 20, 128, 64, "pool", True),
    (3, 50, 32, 128, 64, None, True), (2, 26, 12, 64, 32, "pool", False), (4, 100, 64, 64, 32, "ext", False),
    (26, 100, 40, 64, 32, "ext", True), (100, 50, 20, 128, 64, "pool", False)])   # (the last two: launches of >= 192 workgroups)
def test_dgrad_with_the_batchnorm_apply_pass_in_its_loader(B, H, W, cin, cout, ep, drop):
    """sept_conv5x5_dgrad_bnapply (blocks 2 / 3 of a network without conv weight gradients): the data-gradient conv that forms
    the gradient of a BatchNorm + ReLU + MaxPool block's pre-activations in its tile loader, against the apply pass
    (sept_bn_relu_pool_backward: baseline_models.py:179-182 + autograd) followed by the conv on the stored tensor -- for the
    plain epilogue and both sums epilogues, tiles that start on odd rows, ragged last tiles and dropped channels."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(cin + H + W)
    pre = (torch.randn(B, H, W, cin, generator=g) * 1.3).bfloat16().cuda()
    gp = torch.randn(B, H // 2, W // 2, cin, generator=g).bfloat16().cuda()
    wt = ops.conv5x5_prep_weights((torch.randn(cin, cout, 5, 5, generator=g) * 0.05).cuda(), 1)
    mean, invstd = (0.2 * torch.randn(cin, generator=g)).cuda(), (0.5 + torch.rand(cin, generator=g)).cuda()
    gamma, beta = (1 + 0.3 * torch.randn(cin, generator=g)).cuda(), (0.2 * torch.randn(cin, generator=g)).cuda()
    gamma[3] = -0.6
    dmask = ((torch.rand(B, cin, generator=g) > 0.2).float() * 1.25).cuda() if drop else None
    assert ops.conv5x5_bnapply_supported(pre, cout, ep is not None)
    # the unfused pair: reduce + apply pass, then the conv on the stored gradient
    dpre, _, _ = ops.bn_relu_pool_backward(gp, pre, mean, invstd, gamma, beta, dmask, 2, need_param_grads=False)
    sums, _, _ = ops.bn_backward_sums(gp, pre, mean, invstd, gamma, beta, dmask, 2, need_param_grads=False)
    e_mean, e_invstd = (0.1 * torch.randn(cout, generator=g)).cuda(), (0.5 + torch.rand(cout, generator=g)).cuda()
    e_gamma, e_beta = (1 + 0.3 * torch.randn(cout, generator=g)).cuda(), (0.2 * torch.randn(cout, generator=g)).cuda()
    e_drop = ((torch.rand(B, cout, generator=g) > 0.2).float() * 1.25).cuda() if drop else None
    if ep == "ext":
        ext = (torch.randn(B, H, W, cout, generator=g) * 1.5).bfloat16().cuda()
        want, wsums = ops.conv5x5_dgrad_bnsums_ext(dpre, wt, ext, e_mean, e_invstd, e_gamma, e_beta, e_drop)
        got, gsums = ops.conv5x5_dgrad_bnapply(pre, gp, sums, mean, invstd, gamma, beta, dmask, wt,
                                               ("ext", ext, e_mean, e_invstd, e_gamma, e_beta, e_drop))
    elif ep == "pool":
        ypool = torch.relu(torch.randn(B, H, W, cout, generator=g)).bfloat16().cuda()
        want, wsums = ops.conv5x5_dgrad_bnsums(dpre, wt, ypool, e_gamma, e_beta, e_drop)
        got, gsums = ops.conv5x5_dgrad_bnapply(pre, gp, sums, mean, invstd, gamma, beta, dmask, wt,
                                               ("pool", ypool, e_gamma, e_beta, e_drop))
    else:
        want, wsums = ops.conv5x5(dpre, wt), None
        got, gsums = ops.conv5x5_dgrad_bnapply(pre, gp, sums, mean, invstd, gamma, beta, dmask, wt, None)
    # the loader evaluates sc (ge - m1 - xhat m2) as sc ge + kb + kc x: single bf16 roundings of the staged gradient may differ
    w32, g32 = want.float(), got.float()
    assert float((g32 - w32).norm() / w32.norm()) < 3e-3
    assert float((g32 - w32).abs().max()) < 2e-2 * float(w32.abs().max())
    if ep == "ext":
        assert torch.equal(got == 0, want == 0) or float(((got == 0) != (want == 0)).float().mean()) < 1e-4   # same mask
    if wsums is not None:
        assert gsums is not None    # (the loader form may run another tile shape: the number of partial columns may differ)
        a = gsums[0][:2 * cout * gsums[1]].view(2 * cout, -1).sum(1)
        b = wsums[0][:2 * cout * wsums[1]].view(2 * cout, -1).sum(1)
        assert torch.allclose(a, b, rtol=5e-3, atol=5e-3 * float(b.abs().max()))
    # and against plain torch: the whole block's backward + transposed conv in fp32
    if ep is None and B <= 3:
        x32 = pre.float().permute(0, 3, 1, 2).requires_grad_(True)
        sc = (gamma * invstd).view(1, -1, 1, 1)
        # BatchNorm with the GIVEN statistics treated as batch statistics is not what autograd would differentiate: the sums
        # (m1, m2) are inputs here, so restate the formula instead
        xh = (x32.detach() - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        y = torch.relu(xh * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1))
        yp, idx = torch.nn.functional.max_pool2d(y, 2, return_indices=True)
        gg = gp.float().permute(0, 3, 1, 2) * (dmask.view(B, -1, 1, 1) if drop else 1.0)
        gg = torch.where(yp > 0, gg, torch.zeros_like(gg))
        ge = torch.nn.functional.max_unpool2d(gg, idx, 2, output_size=(H, W))
        n = float(B * H * W)
        m1, m2 = (sums[:cin] / n).view(1, -1, 1, 1), (sums[cin:] / n).view(1, -1, 1, 1)
        d = (sc * (ge - m1 - xh * m2)).bfloat16().float()
        wfull = wt.float()   # [25][cout][cin], taps already flipped for the data gradient
        wconv = wfull.view(5, 5, cout, cin).permute(2, 3, 0, 1).contiguous()
        ref = torch.nn.functional.conv2d(d, wconv, padding=2).permute(0, 2, 3, 1)
        assert float((g32 - ref).norm() / ref.norm()) < 6e-3


@pytest.mark.parametrize("B,H,W,stats,drop", [(3, 100, 40, True, True), (2, 100, 64, True, False), (5, 50, 20, False, True),
                                              (2, 26, 12, True, True), (1, 7, 5, False, False), (26, 100, 40, True, True)])
def test_forward_conv_with_the_pool_first_activation_in_its_loader(B, H, W, stats, drop):
    """sept_conv5x5_forward_act (conv.5 behind a pool-first block 1 of a network without conv weight gradients): the
    activation dropscale * relu(bn(ext)) formed in the conv's tile loader, against sept_bn_relu_ext_forward followed by the
    conv on the stored activation (baseline_models.py:173-178) -- same bits, including the zero padding of the ACTIVATION
    (not of ext) at the image border and the statistics of the output."""
    from sept_amd import ops
    cin, cout = 32, 64
    g = torch.Generator().manual_seed(H * 7 + W)
    ext = (torch.randn(B, H, W, cin, generator=g) * 1.5).bfloat16().cuda()
    mean, invstd = (0.2 * torch.randn(cin, generator=g)).cuda(), (0.5 + torch.rand(cin, generator=g)).cuda()
    gamma, beta = (1 + 0.3 * torch.randn(cin, generator=g)).cuda(), (0.5 + 0.2 * torch.randn(cin, generator=g)).cuda()
    gamma[5] = -0.8
    dmask = ((torch.rand(B, cin, generator=g) > 0.2).float() * 1.25).cuda() if drop else None
    wt = ops.conv5x5_prep_weights((torch.randn(cout, cin, 5, 5, generator=g) * 0.05).cuda(), 0)
    bias = (0.1 * torch.randn(cout, generator=g)).cuda()
    act = ops.bn_relu_ext_forward(ext, None, mean, invstd, gamma, beta, dmask)
    assert float((act[:, 0] != 0).float().mean()) > 0.3          # the border rows carry non-zero activations: padding matters
    if stats:
        want = ops.conv5x5_forward_stats(act, wt, bias)
        got = ops.conv5x5_forward_act(ext, mean, invstd, gamma, beta, dmask, wt, bias, True)
        assert want is not None and got is not None
        for a_, b_ in zip(got, want):
            assert torch.equal(a_, b_)
    else:
        want = ops.conv5x5(act, wt, bias)
        got = ops.conv5x5_forward_act(ext, mean, invstd, gamma, beta, dmask, wt, bias, False)
        assert got is not None and torch.equal(got, want)
