"""GPU parity of the bf16-MFMA 5x5 convolution (forward and data-gradient form) against a
plain fp32 torch conv on the same bf16-rounded operands (reference layers:
model/baseline_models.py:171-189).  Tolerance: fp32 accumulation of bf16 products is exact
up to summation order; the only rounding is the bf16 output (rel 2^-8)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [  # (B, H, W, cin, cout)
    (3, 100, 40, 32, 64),    # conv2 @ 80 mels
    (2, 100, 64, 32, 64),    # conv2 @ 128 mels
    (3, 50, 20, 64, 128),    # conv3 @ 80 mels
    (2, 50, 32, 64, 128),    # conv3 @ 128 mels
    (2, 25, 10, 128, 128),   # deep model 4th conv
    (1, 7, 9, 32, 64),       # ragged: tile tail + tiny image
    (2, 33, 5, 64, 128),
    # launches of >= 192 workgroups: the 512-pixel tile shapes the training step runs (smaller launches take half-size tiles)
    (26, 100, 40, 32, 64), (100, 50, 20, 64, 128),
]


def _ref(x_nhwc, w, bias):
    y = F.conv2d(x_nhwc.float().permute(0, 3, 1, 2), w.bfloat16().float(), bias, padding=2)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,cin,cout", SHAPES)
def test_conv_forward(B, H, W, cin, cout):
    from sept_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, cin, generator=g).bfloat16().cuda()
    w = (torch.randn(cout, cin, 5, 5, generator=g) / (cin * 25) ** 0.5).cuda()
    bias = (0.1 * torch.randn(cout, generator=g)).cuda()
    y = ops.conv5x5(x, ops.conv5x5_prep_weights(w, 0), bias).float()
    want = _ref(x, w, bias)
    assert torch.allclose(y, want, rtol=1e-2, atol=1e-2), (y - want).abs().max()
    # exactness before the final rounding: error bounded by one bf16 ulp of the value
    assert ((y - want).abs() <= want.abs() * 2 ** -7 + 1e-3).all()


@pytest.mark.parametrize("B,H,W,cin,cout", SHAPES[:4] + SHAPES[5:])
def test_conv_data_gradient(B, H, W, cin, cout):
    """dX = conv(dY, flipped/transposed W): compare with autograd of the fp32 conv."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(7 + W)
    dy = torch.randn(B, H, W, cout, generator=g).bfloat16().cuda()
    w = (torch.randn(cout, cin, 5, 5, generator=g) / (cout * 25) ** 0.5).cuda()
    dx = ops.conv5x5(dy, ops.conv5x5_prep_weights(w, 1)).float()
    x = torch.zeros(B, cin, H, W, device="cuda", requires_grad=True)
    F.conv2d(x, w.bfloat16().float(), None, padding=2).backward(dy.float().permute(0, 3, 1, 2))
    want = x.grad.permute(0, 2, 3, 1)
    assert torch.allclose(dx, want, rtol=1e-2, atol=1e-2), (dx - want).abs().max()


def test_conv_forward_with_statistics_random_shapes():
    """Seeded sweep over image sizes that hit every tile-tail case (H*W not a multiple of the 128 / 256-pixel
    tiles, widths down to the 5-pixel minimum, batch sizes around the 8-image XCD groups): output against the
    fp32 conv, statistics against the mean / variance of that output."""
    import random
    from sept_amd import ops
    rnd = random.Random(20)
    for _ in range(14):
        cin, cout = rnd.choice([(32, 64), (64, 128), (128, 128)])
        B, H, W = rnd.randint(1, 11), rnd.randint(5, 70), rnd.randint(5, 44)
        g = torch.Generator().manual_seed(B * 131 + H * 7 + W)
        x = torch.randn(B, H, W, cin, generator=g).bfloat16().cuda()
        w = (torch.randn(cout, cin, 5, 5, generator=g) / (cin * 25) ** 0.5).cuda()
        bias = (0.3 * torch.randn(cout, generator=g)).cuda()
        res = ops.conv5x5_forward_stats(x, ops.conv5x5_prep_weights(w, 0), bias)
        assert res is not None, (B, H, W, cin, cout)
        y, mean, invstd = res
        want = _ref(x, w, bias)
        assert ((y.float() - want).abs() <= want.abs() * 2 ** -7 + 1e-3).all(), (B, H, W, cin, cout)
        yf = y.float().reshape(-1, cout).double()
        assert torch.allclose(mean.double(), yf.mean(0), rtol=1e-5, atol=1e-6), (B, H, W, cin, cout)
        assert torch.allclose(invstd.double(), (yf.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4), (B, H, W, cin, cout)


def test_conv_errors():
    from sept_amd import ops
    from sept_amd._lib import SeptError
    x = torch.zeros(1, 8, 8, 48, dtype=torch.bfloat16, device="cuda")
    wt = torch.zeros(25, 64, 48, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(SeptError):
        ops.conv5x5(x, wt)


@pytest.mark.parametrize("B,H,W,cin,cout", [(3, 100, 40, 32, 64), (2, 100, 64, 32, 64), (3, 50, 20, 64, 128),
                                            (2, 50, 32, 64, 128), (2, 25, 10, 128, 128), (1, 7, 9, 32, 64),
                                            (5, 33, 5, 64, 128)])
def test_conv_weight_gradient(B, H, W, cin, cout):
    """dW through the transposing-LDS-read MFMA kernel vs fp32 autograd on the CPU."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(11 + W + cin)
    x = torch.randn(B, H, W, cin, generator=g).bfloat16()
    dy = torch.randn(B, H, W, cout, generator=g).bfloat16()
    dw = ops.conv5x5_backward_weight(x.cuda(), dy.cuda()).cpu()
    w = torch.zeros(cout, cin, 5, 5, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), w, None, padding=2).backward(dy.float().permute(0, 3, 1, 2))
    scale = w.grad.abs().max()
    assert torch.allclose(dw, w.grad, rtol=1e-3, atol=1e-4 * scale), (dw - w.grad).abs().max() / scale
    # deterministic: same inputs, bit-identical result
    assert torch.equal(dw, ops.conv5x5_backward_weight(x.cuda(), dy.cuda()).cpu())


@pytest.mark.parametrize("B,H,W,cin,cout", [(9, 100, 40, 64, 32), (9, 50, 20, 128, 64), (3, 100, 64, 64, 32), (2, 51, 21, 128, 64),
                                           (26, 100, 40, 64, 32), (100, 50, 20, 128, 64)])   # (the last two: 512-pixel tiles)
@pytest.mark.parametrize("drop", [False, True])
def test_dgrad_conv_with_batchnorm_backward_sums(B, H, W, cin, cout, drop):
    """sept_conv5x5_dgrad_bnsums + sept_bn_relu_pool_backward_presummed against the separate data-gradient conv and
    sept_bn_relu_pool_backward (pooled sums): identical dx of the conv, identical gradient of the pre-activations, the
    same dgamma / dbeta up to summation order -- with one channel chunk that has to take the window path (tiny gamma)."""
    from sept_amd import ops
    g = torch.Generator().manual_seed(cin + H)
    dy2 = (torch.randn(B, H, W, cin, generator=g) * 0.1).bfloat16().cuda()          # gradient of the NEXT conv's output
    w = (torch.randn(cin, cout, 5, 5, generator=g) * 0.05).cuda()                   # that conv's weight (cin here = its cout)
    wtd = ops.conv5x5_prep_weights(w, 1)
    # the block in front: pre-activations x1 (B, 2H, 2W, cout), statistics, pooled output y1 (B, H, W, cout)
    x1 = (torch.randn(B, 2 * H, 2 * W, cout, generator=g) * 1.3 + 0.2).bfloat16().cuda()
    gamma = 1 + 0.3 * torch.randn(cout, generator=g)
    gamma[5], gamma[cout - 3] = 2e-5, -0.6
    gamma, beta = gamma.cuda(), (0.2 * torch.randn(cout, generator=g)).cuda()
    dmask = ((torch.rand(B, cout, generator=g) > 0.2).float() * 1.25).cuda() if drop else None
    mean, invstd = ops.bn_stats(x1)
    y1 = ops.bn_relu_pool_forward(x1, mean, invstd, gamma, beta, dmask, 2)
    want_dx = ops.conv5x5(dy2, wtd)
    want = ops.bn_relu_pool_backward(want_dx, x1, mean, invstd, gamma, beta, dmask, 2, y=y1)
    dx, presums = ops.conv5x5_dgrad_bnsums(dy2, wtd, y1, gamma, beta, dmask)
    assert torch.equal(dx, want_dx)
    if presums is None:      # no epilogue form for this shape: the caller falls back to the separate pass
        assert ops.lib.sept_conv5x5_bwsums_parts(B, H, W, cin, cout) == 0
        return
    got = ops.bn_relu_pool_backward_presummed(dx, x1, mean, invstd, gamma, beta, dmask, presums, 2)
    for a_, b_ in ((got[1], want[1]), (got[2], want[2])):
        assert torch.allclose(a_, b_, rtol=2e-4, atol=2e-4 * float(b_.abs().max())), (a_ - b_).abs().max()
    d, d2 = want[0].float(), got[0].float()
    assert float((d - d2).norm() / d.norm()) < 1e-3           # the sums enter every element through the mean terms
    assert ((d - d2).abs() <= d.abs() * 2 ** -6 + 2e-3 * float(d.abs().max())).all()
