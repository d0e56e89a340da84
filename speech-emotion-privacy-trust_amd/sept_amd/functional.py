"""The model half of the hot path as compositions of libsept_hip kernels.

`trunk_forward` / `trunk_backward` run one two_d_cnn_lstm-style network (conv stack ->
bi-GRU -> pooling -> dense1 -> head; reference model/baseline_models.py:222-260 and the
re-statements of it inside the cloak wrappers, model/cloak_models.py:165-224) on NHWC bf16
activations; the autograd Functions below glue them into torch's autograd graph so the
product modules (model/*.py) keep the reference's call surface.  Nothing here computes on the
CPU or through eager torch math: torch supplies storage, views, RNG draws and the autograd
tape only.
"""
import math
import os
from types import SimpleNamespace

import torch

from . import ops
from ._lib import SeptError, require_cuda

DROP_P = 0.2  # every Dropout / Dropout2d / GRU dropout in the reference uses 0.2 (baseline_models.py:153)

# ---------------------------------------------------------------------------------------------
# derived-operand cache (bf16 conv operands, permuted GRU weights).  Keyed on storage + version;
# the HIP optimiser updates parameters through raw pointers, so it bumps EPOCH explicitly.
# ---------------------------------------------------------------------------------------------
_EPOCH = [0]


def invalidate_weight_cache():
    _EPOCH[0] += 1


def _stamp(t, t2=None):
    # frozen parameters (requires_grad False) are never touched by the raw-pointer optimiser
    if t2 is None:
        return (t.data_ptr(), t._version, _EPOCH[0] if t.requires_grad else -1)
    return (t.data_ptr(), t._version, t2.data_ptr(), t2._version, _EPOCH[0] if (t.requires_grad or t2.requires_grad) else -1)


def _cached(tag, t, fn):
    """Derived operand of parameter `t`, stored ON the parameter object (so it can never be
    served to another tensor that happens to reuse the address) and refreshed whenever torch
    bumps the tensor version or the HIP optimiser bumps the epoch."""
    store = t.__dict__.setdefault("_sept_derived", {})
    stamp = _stamp(t)
    hit = store.get(tag)
    if hit is None or hit[0] != stamp:
        hit = store[tag] = (stamp, fn())
    return hit[1]


def _cached_pair(tag, t1, t2, fn):
    """As _cached, for an operand derived from two parameters (forward + reverse GRU weights)."""
    store = t1.__dict__.setdefault("_sept_derived", {})
    stamp = _stamp(t1, t2)
    hit = store.get(tag)
    if hit is None or hit[0] != stamp:
        hit = store[tag] = (stamp, fn())
    return hit[1]


def _is_stale(tag, t, t2=None):
    hit = t.__dict__.get("_sept_derived", {}).get(tag)
    return hit is None or hit[0] != _stamp(t, t2)


def _store(tag, t, value, t2=None):
    t.__dict__.setdefault("_sept_derived", {})[tag] = (_stamp(t, t2), value)


def prepare_operands(P, W_in, need_dgrad=True):
    """Every weight-only operand of network P that is out of date -- conv1's operand block, the bf16 operands of the 5x5 convs
    (forward and, when a backward pass will follow, data-gradient orientation), the packed recurrent input matrices -- built
    by ONE launch (ops.prepare_operands) and placed in the cache entries trunk_forward / trunk_backward look up.  A trainable
    network's weights change every optimiser step, so without this its forward chain opens seven consumers with a 5 us
    operand-build launch each (the frozen network builds them once for good).  One stale operand is left to its consumer."""
    items, sinks = [], []
    fp32 = lambda t: t.dtype == torch.float32 and t.is_contiguous()   # noqa: E731  (parameters: views of the flat buffer)
    for li, cv in enumerate(P.convs):
        w = cv.weight
        if not fp32(w):
            continue
        if li == 0:
            if w.shape != (32, 1, 5, 5) or (cv.bias is not None and not fp32(cv.bias)):
                continue
            if _is_stale("conv1prep", w, cv.bias):
                items.append(("conv1", w.detach(), None if cv.bias is None else cv.bias.detach()))
                sinks.append(lambda v, w=w, b=cv.bias: _store("conv1prep", w, v, b))
        else:
            for tag, mode in (("convfwd", 0), ("convdgrad", 1)):
                if (mode == 0 or need_dgrad) and _is_stale(tag, w):
                    items.append(("conv5x5", w.detach(), mode))
                    sinks.append(lambda v, w=w, tag=tag: _store(tag, w, v))
    wd = W_in
    for pool in P.pools:
        wd //= pool
    C = P.convs[-1].weight.shape[0]
    r = P.rnn
    for layer in range(2):
        wif, wir = getattr(r, f"weight_ih_l{layer}"), getattr(r, f"weight_ih_l{layer}_reverse")
        bif, bir = getattr(r, f"bias_ih_l{layer}"), getattr(r, f"bias_ih_l{layer}_reverse")
        if all(fp32(t) for t in (wif, wir, bif, bir)) and _is_stale(f"wih_cat{layer}", wif, wir):
            items.append(("gru", wif.detach(), wir.detach(), bif.detach(), bir.detach(), C if layer == 0 else 0, wd if layer == 0 else 0))
            sinks.append(lambda v, wif=wif, wir=wir, layer=layer: _store(f"wih_cat{layer}", wif, v, wir))
    if len(items) < 2:
        return 0
    for k in range(0, len(items), 12):
        for sink, v in zip(sinks[k:k + 12], ops.prepare_operands(items[k:k + 12], P.convs[0].weight.device)):
            sink(v)
    return len(items)


def _gru_cat_weights(wif, wir, bif, bir, layer, C, Wd):
    """[W_ih forward; W_ih reverse] as ONE (2G, K) operand (G = 3 * hidden; layer 0: columns permuted to
    the NHWC feature order), its (K, 2G) transpose (the k-contiguous operand of dx = dgi W) and the
    matching (2G,) bias, so both directions share one product per GEMM.  One launch (sept_gru_pack)."""
    G, K = wif.shape
    dev = wif.device
    wcat = torch.empty((2 * G, K), dtype=torch.float32, device=dev)
    wcatT = torch.empty((K, 2 * G), dtype=torch.float32, device=dev)
    bcat = torch.empty(2 * G, dtype=torch.float32, device=dev)
    ops.check(ops.lib.sept_gru_pack(wif.data_ptr(), wir.data_ptr(), bif.data_ptr(), bir.data_ptr(), G, K,
                                    C if layer == 0 else 0, Wd if layer == 0 else 0, wcat.data_ptr(), wcatT.data_ptr(),
                                    bcat.data_ptr(), ops.current_stream_ptr(dev)), "sept_gru_pack")
    return wcat, bcat, wcatT


def _conv1_operand(cv):
    """conv1's weights + bias in operand form, rebuilt only when either parameter changed (one launch per optimiser
    step for a trainable layer, one for good for a frozen one -- round 1 rebuilt it in every conv1 call, 5 per step)"""
    if cv.bias is None:
        return _cached("conv1prep", cv.weight, lambda: ops.conv1_prep(cv.weight, None))
    return _cached_pair("conv1prep", cv.weight, cv.bias, lambda: ops.conv1_prep(cv.weight, cv.bias))


def _nt_ok(K):
    return K % 32 == 0


# ---------------------------------------------------------------------------------------------
# parameter view of a two_d_cnn_lstm / deep_two_d_cnn_lstm module
# ---------------------------------------------------------------------------------------------
def trunk_params(model, head: str, att="model"):
    """Collect the tensors the HIP trunk reads from a (reference-layout) module.  `head` is
    'emotion', 'gender' or 'multitask' (which prediction layer(s) the caller applies); `att` is the
    attention mode ('model' = the module's own `att`; the cloak wrappers pass the emotion model's
    setting for both branches, as cloak_models.py:171/208 do)."""
    conv = model.conv
    if len(conv) == 2 and not isinstance(conv[0], torch.nn.Conv2d):  # Sequential(GradientReversal, conv)
        conv = conv[1]
    convs = [m for m in conv if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in conv if isinstance(m, torch.nn.BatchNorm2d)]
    pools, drop_ps = [], []
    mods = list(conv)
    for i, m in enumerate(mods):
        if isinstance(m, torch.nn.Conv2d):
            nxt = []
            for n in mods[i + 1:]:
                if isinstance(n, torch.nn.Conv2d):
                    break
                nxt.append(n)
            pools.append(2 if any(isinstance(n, torch.nn.MaxPool2d) for n in nxt) else 1)
            dps = [n.p for n in nxt if isinstance(n, torch.nn.Dropout2d)]
            drop_ps.append(dps[0] if dps else 0.0)
    rnn = model.rnn
    if not isinstance(rnn, (torch.nn.GRU, torch.nn.LSTM)):
        raise NotImplementedError("the HIP path implements rnn_cell 'gru' and 'lstm'")
    if rnn.hidden_size not in (64, 128) or rnn.num_layers != 2 or not rnn.bidirectional:
        raise NotImplementedError(
            f"rnn hidden_size={rnn.hidden_size} num_layers={rnn.num_layers} bidirectional={rnn.bidirectional}: the HIP "
            "recurrences (sept_gru_* / sept_lstm_*) support hidden 64 (the trainers' config) or 128 (the class "
            "default) with 2 bidirectional layers; there is no fallback for other shapes")
    att = model.att if att == "model" else att
    if att not in (None, "self_att"):
        raise ValueError(f"unknown attention mode {att!r}")
    heads = {"emotion": [model.pred_emotion_layer], "gender": [model.pred_gender_layer],
             "multitask": [model.pred_emotion_layer, model.pred_gender_layer]}[head]
    return SimpleNamespace(convs=convs, bns=bns, pools=pools, drop_ps=drop_ps, dense_p=model.dropout.p, rnn=rnn,
                           dense1=model.dense1, heads=heads, att=att,
                           att1=model.att_linear1 if att else None, att2=model.att_linear2 if att else None,
                           training=model.training)


def _drop_mask(shape, device, p=DROP_P):
    """0 / (1/(1-p)) scale mask from the device's Philox stream (one HIP launch)."""
    return ops.rng(device, "dropout").dropout_mask(shape, p)


# sync-BN (option 1): when enabled, BatchNorm statistics and the two backward
# sums are all-reduced over the process group, so a data-parallel step sees the statistics of the
# global batch.  Off by default (per-rank statistics, option 2).
_SYNC_BN = {"on": False, "group": None}


def set_sync_bn(enabled: bool, group=None):
    _SYNC_BN["on"] = bool(enabled) and torch.distributed.is_available() and torch.distributed.is_initialized()
    _SYNC_BN["group"] = group


# Philox call-site ids of the two networks of the GRL step: their masks must not depend on which branch the host
# enqueues first (the hand-scheduled step enqueues the gender chain first, the autograd path the emotion one)
SITE_EMOTION, SITE_GENDER = 1000, 2000


def _drop_masks(device, specs, site=None):
    """All dropout scale masks of one forward in one launch per distinct p: `specs` is a list of
    (key, shape, p); returns {key: mask} for the entries with p > 0.  `site`: first explicit Philox call-site id
    (one per distinct p), None = the stream's sequential numbering."""
    out = {}
    by_p = {}
    for key, shape, p in specs:
        if p > 0:
            by_p.setdefault(float(p), []).append((key, shape))
    for k, (p, items) in enumerate(by_p.items()):
        sizes = [(math.prod(shape) + 7) // 8 * 8 for _, shape in items]
        flat = ops.rng(device, "dropout").dropout_mask((sum(sizes),), p, site=None if site is None else site + k)
        off = 0
        for (key, shape), n in zip(items, sizes):
            out[key] = flat[off:off + math.prod(shape)].view(shape)
            off += n
    return out


def step_masks(P, B, H, device, inj=None, site=None):
    """Every dropout mask one training-mode forward of network P needs at batch B and input height H, drawn up front in
    one launch per distinct p (keys: ('c', layer), 'rnn', 'dense'); entries covered by `inj` (explicit test masks) are left
    out."""
    inj = inj or {}
    t_out = H
    for pool in P.pools:
        t_out //= pool
    specs = [(("c", li), (B, cv.weight.shape[0]), P.drop_ps[li]) for li, cv in enumerate(P.convs)
             if "drop2d" not in inj]
    if "rnn" not in inj:
        specs.append(("rnn", (B, t_out, 2 * P.rnn.hidden_size), P.rnn.dropout))
    if "dense" not in inj:
        specs.append(("dense", (B, P.dense1.weight.shape[0]), P.dense_p))
    return _drop_masks(device, specs, site)


def trunk_forward(x, P, pooling="mean", need_grad=True, injected=None, gfeat=None, masks=None, rng_site=None):
    """x (B, H, W) fp32 CUDA -> logits (B, C).  Returns (logits, saved) where `saved` holds
    what trunk_backward needs.  BatchNorm uses batch statistics (and updates the running
    buffers) when the module is in train mode -- also for a frozen model;
    Dropout / Dropout2d / GRU dropout are active in train mode.  `injected` (or `P.injected`, which the cloak wrappers'
    `injected_masks` test hook sets) may carry explicit dropout SCALE masks {'drop2d': [(B, C) per conv block], 'rnn':
    (B, T, 2H), 'dense': (B, 128)} (0 or 1/(1-p)) for reproducible tests."""
    require_cuda(x)
    B, H, W = x.shape
    dev = x.device
    train = P.training
    prepare_operands(P, W, need_dgrad=need_grad)
    inj = injected or getattr(P, "injected", None) or {}
    if inj:
        inj = {k: ([m.to(dev, torch.float32).contiguous() for m in v] if k == "drop2d" else v.to(dev, torch.float32).contiguous())
               for k, v in inj.items()}
    S = SimpleNamespace(x=x, blocks=[], train=train, pooling=pooling, B=B)
    act = None
    h, w = H, W
    if masks is None:   # every mask the step needs, drawn up front (injected ones take precedence below)
        masks = step_masks(P, B, H, dev, inj, rng_site) if train else {}
    for li, (cv, bn, pool) in enumerate(zip(P.convs, P.bns, P.pools)):
        cout = cv.weight.shape[0]
        if li == 0 and pool == 2 and not bn.training and not need_grad and ops.conv1_fused_supported(H, W):
            # inference: block 1 in ONE pass with the running statistics -- conv1 -> BatchNorm -> ReLU -> pool (-> dropout
            # scale) in registers, no 32-channel pre-activation tensor (67 us where conv1 + BatchNorm take 57 + 44 at 224
            # windows).  (Round 2 also ran this form in training, by recomputation; it lost to streaming the tensor and then
            # to the pool-first form below: DESIGN.md section 8.)
            c1p = _conv1_operand(cv)
            mean, invstd = ops.bn_eval_stats(bn.running_mean, bn.running_var, bn.eps)
            drop = None
            if train and (P.drop_ps[li] > 0 or "drop2d" in inj):
                d2 = inj.get("drop2d")
                drop = d2[li] if d2 is not None else masks[("c", li)]
            out = ops.conv1_bn_relu_pool_forward(x, cv.weight, cv.bias, mean, invstd, bn.weight, bn.bias, drop, prep=c1p)
            S.blocks.append(SimpleNamespace(inp=None, pre=None, ext=None, out=out, mean=mean, invstd=invstd, drop=drop, pool=pool,
                                            h=h, w=w, bn_train=False, sync=False, idx=None))
            act = out
            h, w = h // pool, w // pool
            continue
        if (li == 0 and pool == 2 and bn.training and not _SYNC_BN["on"] and L1_POOL_FIRST
                and ops.conv1_pool_supported(H, W, backward=need_grad) and _pool_first_backward_ok(cv, need_grad)):
            # block 1 in pool-first form: conv1 leaves the statistics partials, the 2x2 window's extremum (by the sign of
            # gamma) and its position; a quarter-size elementwise pass forms the pooled activation.  Nothing of
            # (B, H, W, 32) is written or read, forward or backward (include/sept.h, "POOL-FIRST")
            ext, idx, mean, invstd = ops.conv1_forward_pool(x, cv.weight, cv.bias, bn.weight, bn.running_mean, bn.running_var,
                                                            bn.num_batches_tracked,
                                                            bn.momentum if bn.momentum is not None else 0.1, bn.eps,
                                                            prep=_conv1_operand(cv))
            drop = None
            if train and (P.drop_ps[li] > 0 or "drop2d" in inj):
                d2 = inj.get("drop2d")
                drop = d2[li] if d2 is not None else masks[("c", li)]
            # the activation pass: inside the NEXT conv's tile loader when nothing else reads the activation tensor (that conv
            # needs no weight gradient: the frozen network) -- resolved when that conv is reached; else a quarter-size pass now
            defer = BN_ACT_IN_CONV and li + 1 < len(P.convs) and not P.convs[li + 1].weight.requires_grad
            out = None if defer else ops.bn_relu_ext_forward(ext, None, mean, invstd, bn.weight, bn.bias, drop)   # idx stays pure positions
            S.blocks.append(SimpleNamespace(inp=None, pre=None, ext=ext, out=out, mean=mean, invstd=invstd, drop=drop, pool=pool,
                                            h=h, w=w, bn_train=True, sync=False, idx=idx))
            act = out
            h, w = h // pool, w // pool
            continue
        fused_stats = li == 0 and bn.training and not _SYNC_BN["on"] and W + 4 <= 512
        if fused_stats:   # conv1 leaves the statistics partials of its output: no BN pass over 64 B/pixel
            pre, mean, invstd = ops.conv1_forward_stats(x, cv.weight, cv.bias, bn.running_mean, bn.running_var,
                                                        bn.num_batches_tracked,
                                                        bn.momentum if bn.momentum is not None else 0.1, bn.eps,
                                                        prep=_conv1_operand(cv))
        elif li == 0:
            pre = ops.conv1_forward(x, cv.weight, cv.bias, prep=_conv1_operand(cv))
        else:
            wt = _cached("convfwd", cv.weight, lambda: ops.conv5x5_prep_weights(cv.weight, 0))
            res = None
            prevb = S.blocks[li - 1]
            if prevb.out is None:   # a pool-first block whose activation pass was deferred into this conv's loader
                pbn = P.bns[li - 1]
                want = CONV_FUSED_STATS and bn.training and not _SYNC_BN["on"]
                r = ops.conv5x5_forward_act(prevb.ext, prevb.mean, prevb.invstd, pbn.weight, pbn.bias, prevb.drop, wt, cv.bias,
                                            want, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                            bn.momentum if bn.momentum is not None else 0.1, bn.eps)
                if r is None:       # no such kernel form for this shape: the activation tensor after all
                    prevb.out = act = ops.bn_relu_ext_forward(prevb.ext, None, prevb.mean, prevb.invstd, pbn.weight, pbn.bias,
                                                              prevb.drop)
                elif want:
                    res = r
                else:
                    res = (r, None, None)
            if res is not None:
                pass
            elif CONV_FUSED_STATS and bn.training and not _SYNC_BN["on"]:   # statistics in the conv's epilogue
                res = ops.conv5x5_forward_stats(act, wt, cv.bias, bn.running_mean, bn.running_var,
                                                bn.num_batches_tracked,
                                                bn.momentum if bn.momentum is not None else 0.1, bn.eps)
            if res is not None:
                pre, mean, invstd = res
                fused_stats = mean is not None
            else:
                pre = ops.conv5x5(act, wt, cv.bias)
        if fused_stats:
            pass
        elif bn.training:
            mean, invstd = ops.bn_stats(pre, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                        bn.momentum if bn.momentum is not None else 0.1, bn.eps,
                                        sync_group=_SYNC_BN["group"], sync=_SYNC_BN["on"])
        else:
            mean, invstd = ops.bn_eval_stats(bn.running_mean, bn.running_var, bn.eps)
        drop = None
        if train and (P.drop_ps[li] > 0 or "drop2d" in inj):
            d2 = inj.get("drop2d")
            drop = d2[li] if d2 is not None else masks[("c", li)]
        idx = None
        if (li == 0 and need_grad and L1_SPARSE and bn.training and pool == 2 and not _SYNC_BN["on"] and h % 2 == 0
                and h >= 4 and w % 8 == 0 and w <= 128 and not cv.weight.requires_grad):
            # block 1's backward pass will work from the pooled gradient + the window positions (conv1_backward_data_sparse)
            out, idx = ops.bn_relu_pool_forward(pre, mean, invstd, bn.weight, bn.bias, drop, pool, want_argmax=True)
        else:
            out = ops.bn_relu_pool_forward(pre, mean, invstd, bn.weight, bn.bias, drop, pool)
        S.blocks.append(SimpleNamespace(inp=act, pre=pre, ext=None, out=out, mean=mean, invstd=invstd, drop=drop, pool=pool, h=h,
                                        w=w, bn_train=bn.training, sync=_SYNC_BN["on"] and bn.training, idx=idx))
        act = out
        h, w = h // pool, w // pool
    # ---- GRU: (B, T=h, D = w*C) with NHWC feature order (w, c) ----
    C = act.shape[-1]
    T, D = h, w * C
    seq = act.view(B * T, D)
    r = P.rnn
    S.seq, S.T, S.D, S.C, S.Wd = seq, T, D, C, w
    Hh = r.hidden_size            # hidden units per direction; H2 = GRU output width, G = 3 gates per direction
    lstm = isinstance(r, torch.nn.LSTM)        # gate blocks per direction: 4 (i, f, g, o) or 3 (r, z, n)
    H2, G = 2 * Hh, (4 if lstm else 3) * Hh
    S.Hh, S.lstm = Hh, lstm
    layer_in = seq
    S.gru = []
    for layer in range(2):
        names = [f"weight_ih_l{layer}", f"weight_ih_l{layer}_reverse", f"weight_hh_l{layer}",
                 f"weight_hh_l{layer}_reverse", f"bias_ih_l{layer}", f"bias_ih_l{layer}_reverse",
                 f"bias_hh_l{layer}", f"bias_hh_l{layer}_reverse"]
        wif, wir, whf, whr, bif, bir, bhf, bhr = [getattr(r, n) for n in names]
        wcat, bcat, wcatT = _cached_pair(f"wih_cat{layer}", wif, wir,
                                         lambda: _gru_cat_weights(wif, wir, bif, bir, layer, C, w))
        K = layer_in.shape[1]
        # (B*T, 2G): both directions in one product, on the split-bf16 matrix pipe when K allows
        gi = ops.linear_nt_split(layer_in, wcat, bcat) if _nt_ok(K) else ops.linear_forward(layer_in, wcat, bcat)
        cells = None
        lmask = None
        if layer == 0 and train and (r.dropout > 0 or "rnn" in inj):
            m = inj.get("rnn")
            lmask = (m if m is not None else masks["rnn"]).contiguous()
        outm = None
        if lstm:
            out, gates, cells = ops.lstm_forward(gi.view(B, T, 2, G), whf, whr, bhf, bhr)
        elif lmask is not None:    # the inter-layer dropout is applied by the recurrence kernel itself (second output)
            out, gates, outm = ops.gru_forward(gi.view(B, T, 2, G), whf, whr, bhf, bhr, mask=lmask.view(B, T, H2))
        else:
            out, gates = ops.gru_forward(gi.view(B, T, 2, G), whf, whr, bhf, bhr)
        Gs = SimpleNamespace(inp=layer_in, out=out, gates=gates, cells=cells, wcat=wcat, wcatT=wcatT, whf=whf, whr=whr,
                             mask=lmask)
        if layer == 0:
            nxt = out.view(B * T, H2)
            if lmask is not None:
                nxt = (outm if outm is not None else ops.mul(out, lmask)).view(B * T, H2)
            layer_in = nxt
        S.gru.append(Gs)
    out1 = S.gru[1].out
    S.att = None
    S.fused_head = (P.att is None and pooling == "mean" and gfeat is None and H2 <= 256
                    and P.dense1.weight.shape == (P.dense1.weight.shape[0], H2) and P.dense1.weight.shape[0] <= 256)
    if S.fused_head:   # mean over time + dense1 + ReLU/dropout + prediction layer(s) in one launch
        dmask = None
        if train and (P.dense_p > 0 or "dense" in inj):
            m = inj.get("dense")
            dmask = m if m is not None else masks["dense"]
        if len(P.heads) == 1:
            wh, bh = P.heads[0].weight, P.heads[0].bias
        else:   # pred='multitask': the two prediction layers stacked into one (NC, D1) operand
            wh = torch.cat([h.weight.detach() for h in P.heads]).contiguous()
            bh = torch.cat([h.bias.detach() for h in P.heads]).contiguous()
        logits, z, d1, d1a = ops.head_forward(out1, P.dense1.weight, P.dense1.bias, dmask, wh, bh)
        S.zdim = z.shape[1]
        S.z, S.d1, S.dmask, S.d1a, S.wh = z, d1, dmask, d1a, wh
        return logits, (S if need_grad else None)
    if P.att == "self_att":   # baseline_models.py:233-242: 16-head additive attention over time
        x2 = out1.view(B * T, H2)
        a1t = ops.tanh_forward(ops.linear_forward(x2, P.att1.weight, P.att1.bias))
        scores = ops.linear_forward(a1t, P.att2.weight, P.att2.bias)
        z, probs = ops.att_pool_forward(scores.view(B, T, -1), out1)
        S.att = SimpleNamespace(a1t=a1t, probs=probs)
    else:
        z = ops.mean_t_forward(out1) if pooling == "mean" else out1.view(B, T * H2)
    S.zdim = z.shape[1]
    if gfeat is not None:     # :244-245: utterance-level functionals appended to the pooled vector
        z = torch.cat((z, gfeat.detach().to(z.dtype).view(B, -1)), 1)
    if z.shape[1] != P.dense1.weight.shape[1]:
        raise SeptError(f"dense1 expects {P.dense1.weight.shape[1]} features, the pooled vector has {z.shape[1]} "
                        "(global_feature given / constructor's global_feature flag / pooling mismatch)")
    d1 = ops.linear_forward(z, P.dense1.weight, P.dense1.bias)
    dmask = None
    if train and (P.dense_p > 0 or "dense" in inj):
        m = inj.get("dense")
        dmask = m if m is not None else masks["dense"]
    d1a = ops.relu_dropout_forward(d1, dmask)
    ncls = [h.weight.shape[0] for h in P.heads]
    logits = torch.empty((B, sum(ncls)), dtype=torch.float32, device=dev)   # heads side by side
    c0 = 0
    for h, n in zip(P.heads, ncls):
        ops.linear_forward(d1a, h.weight, h.bias, out=logits[:, c0:c0 + n])
        c0 += n
    S.z, S.d1, S.dmask, S.d1a = z, d1, dmask, d1a
    return logits, (S if need_grad else None)


def grad_out(param):
    """Where a weight gradient should be WRITTEN: a fresh view of the parameter's slot in its trainer's flat
    gradient buffer (FlatParams registers `_sept_flat` on the parameter), or None (the op allocates).  A fresh
    view object per call, so autograd's AccumulateGrad can adopt it as .grad without a copy."""
    slot = getattr(param, "_sept_flat", None)
    if slot is None or not param.requires_grad:
        return None
    fp, off, n = slot
    return fp.grad[off:off + n].view(param.shape)


def grad_out_pair(p_fwd, p_rev):
    """One (2 x rows, ...) view over the slots of a forward / reverse GRU parameter pair when the trainer packed them
    next to each other (FlatParams does): the product that yields both directions' gradients writes them in place."""
    a, b = getattr(p_fwd, "_sept_flat", None), getattr(p_rev, "_sept_flat", None)
    if a is None or b is None or a[0] is not b[0] or b[1] != a[1] + a[2] or not (p_fwd.requires_grad and p_rev.requires_grad):
        return None
    fp, off, n = a
    return fp.grad[off:off + 2 * n].view((2 * p_fwd.shape[0],) + tuple(p_fwd.shape[1:]))


def zero_bias_grad(bias):
    """The gradient of a conv bias that feeds straight into a training-mode BatchNorm: the BatchNorm's backward has zero
    channel sum by construction, so d(bias) == 0 exactly (the reference gets fp32 summation noise around it).  Returns the
    parameter's slot in the flat gradient buffer -- zero since the buffer was packed, written by nobody else, so no launch
    is needed after the first one -- or a fresh zero tensor for a parameter without a slot."""
    slot = getattr(bias, "_sept_flat", None)
    if slot is None or not bias.requires_grad:
        return ops.fill(torch.empty_like(bias), 0.0)
    fp, off, n = slot
    view = fp.grad[off:off + n].view(bias.shape)
    clean = fp.__dict__.setdefault("_zero_slots", set())
    if off not in clean:     # also under a capture: the slot was zeroed when it was first used and nothing ever writes it
        ops.fill(view, 0.0)
        clean.add(off)
    return view


def _into(out, src):
    """src copied into the flat slot `out` when there is one"""
    return src if out is None else ops.copy_into(out, src.contiguous())


def head_ce_fusable(S, P):
    """the head's backward kernel can form the cross-entropy gradient itself (ops.head_backward_ce): fused head, ONE
    prediction layer of at most 8 classes"""
    return bool(getattr(S, "fused_head", False)) and len(P.heads) == 1 and P.heads[0].weight.shape[0] <= 8


def trunk_backward(S, P, dlogits, need_wgrad=True, need_dx=True, sum_dx=False, ce=None):
    """Returns (dx (B,H,W) fp32 or None, grads: dict parameter-tensor-id -> gradient) for one
    network.  With need_wgrad False (frozen model) only the data path is evaluated.  Weight gradients are
    written straight into the flat gradient buffer of the trainer that owns the parameters (grad_out).
    `ce` = (logits, labels, weights, scale) instead of `dlogits` (head_ce_fusable(S, P) only): the gradient of the weighted
    cross-entropy is formed inside the head's backward kernel, so no loss kernel stands on the chain in front of it.
    sum_dx: the caller only needs the input gradient SUMMED over the batch (the cloak's backward pass: its parameters are
    shared by every sample) -- dx may then come back as (1,H,W), formed without any per-sample pass (pool-first block 1)."""
    B, T = S.B, S.T
    grads = {}
    gout = grad_out if need_wgrad else (lambda p: None)

    def put(param, g):
        if need_wgrad and param is not None and param.requires_grad:
            grads[param] = g

    if ce is not None:
        if not head_ce_fusable(S, P):
            raise SeptError("trunk_backward(ce=...): this head does not take the fused cross-entropy gradient")
    else:
        dlogits = dlogits.contiguous()
    sq = _SideQueue(S.d1.device if ce is not None else dlogits.device, need_wgrad)
    Hh = S.Hh
    H2, G = 2 * Hh, (4 if S.lstm else 3) * Hh
    if S.fused_head:
        if ce is not None:
            dlogits, d_d1, dout = ops.head_backward_ce(ce[0], ce[1], ce[2], ce[3], S.wh, S.d1, S.dmask, P.dense1.weight, T)
        else:
            d_d1, dout = ops.head_backward(dlogits, S.wh, S.d1, S.dmask, P.dense1.weight, T)
        if need_wgrad:
            def head_wgrads():
                c0 = 0
                for h in P.heads:
                    n = h.weight.shape[0]
                    dl = dlogits[:, c0:c0 + n]
                    c0 += n
                    put(h.weight, ops.linear_backward_weight(dl, S.d1a, out=gout(h.weight)))
                    put(h.bias, ops.colsum(dl, out=gout(h.bias)))
                put(P.dense1.weight, ops.linear_backward_weight(d_d1, S.z, out=gout(P.dense1.weight)))
                put(P.dense1.bias, ops.colsum(d_d1, out=gout(P.dense1.bias)))
            sq.small(head_wgrads, dlogits, d_d1, S.d1a, S.z)
        return _trunk_backward_rnn_conv(S, P, dout, grads, put, need_wgrad, need_dx, sq, gout, sum_dx)
    d_d1a, c0 = None, 0
    for h in P.heads:
        n = h.weight.shape[0]
        dl = dlogits[:, c0:c0 + n]
        c0 += n
        if d_d1a is None:
            d_d1a = ops.linear_backward_input(dl, h.weight)
        else:   # second head of pred='multitask': accumulate
            ops.gemm_raw(dl, dl.stride(0), 1, h.weight, h.weight.shape[1], 1, d_d1a, d_d1a.stride(0), B,
                         h.weight.shape[1], n, beta=1.0)
        if need_wgrad:
            put(h.weight, ops.linear_backward_weight(dl, S.d1a, out=gout(h.weight)))
            put(h.bias, ops.colsum(dl, out=gout(h.bias)))
    d_d1 = ops.relu_dropout_backward(d_d1a, S.d1, S.dmask)
    dz = ops.linear_backward_input(d_d1, P.dense1.weight)
    if need_wgrad:
        put(P.dense1.weight, ops.linear_backward_weight(d_d1, S.z, out=gout(P.dense1.weight)))
        put(P.dense1.bias, ops.colsum(d_d1, out=gout(P.dense1.bias)))
    if dz.shape[1] != S.zdim:   # the appended global features are inputs: no gradient needed
        dz = dz[:, :S.zdim].contiguous()
    Hh = S.Hh
    H2, G = 2 * Hh, (4 if S.lstm else 3) * Hh
    if S.att is not None:
        out1 = S.gru[1].out
        x2 = out1.view(B * T, H2)
        dout, dscores = ops.att_pool_backward(dz, out1, S.att.probs)
        ds2 = dscores.view(B * T, -1)
        d_a1 = ops.tanh_backward(ops.linear_backward_input(ds2, P.att2.weight), S.att.a1t)
        if need_wgrad:
            put(P.att2.weight, ops.linear_backward_weight(ds2, S.att.a1t, out=gout(P.att2.weight)))
            put(P.att1.weight, ops.linear_backward_weight(d_a1, x2, out=gout(P.att1.weight)))
            if P.att2.bias is not None:   # deep variant: Linear with bias
                put(P.att2.bias, ops.colsum(ds2, out=gout(P.att2.bias)))
            if P.att1.bias is not None:
                put(P.att1.bias, ops.colsum(d_a1, out=gout(P.att1.bias)))
        w1 = P.att1.weight
        ops.gemm_raw(d_a1, d_a1.stride(0), 1, w1, w1.shape[1], 1, dout, H2, B * T, H2, w1.shape[0], beta=1.0)
    else:
        dout = ops.mean_t_backward(dz, T) if S.pooling == "mean" else dz.view(B, T, H2)
    return _trunk_backward_rnn_conv(S, P, dout, grads, put, need_wgrad, need_dx, sq, gout, sum_dx)


# Weight gradients are off the critical path of the backward pass (nothing downstream reads them until
# the optimiser), so they can be enqueued on a side stream while the data-gradient chain continues
# (SEPT_WGRAD_STREAM=0 keeps one stream).  Which ones go there, and where the side stream joins, depends
# on the stream the network runs on:
#   * a top-level stream (the baseline trainer's single model): all of them; the join is the last thing
#     the network's backward does, so autograd and any caller see finished gradients -- safe for a plain
#     `loss.backward()` of the drop-in modules;
#   * one of the two GRL branch streams (registered in NO_WGRAD_FORK; themselves forked from the caller's
#     stream): a forked stream joining another forked stream inside a HIP-graph capture crashes
#     hipStreamEndCapture on ROCm 7.2 (fork -> fork is fine, the nested JOIN is not), so the side stream is
#     only used under `backward(loss)` below -- the trainers' entry -- which joins it into the caller's
#     stream right after autograd returns and before anything reads a gradient; a bare `loss.backward()`
#     keeps everything in line.  And only the SMALL products go there (head / dense / recurrent weight and
#     bias gradients: ~25 latency-bound launches that otherwise sit in the branch's critical chain); the conv
#     weight gradients stay in line -- as a third queue of MFMA work they made the step slower (2.74 -> 3.00 ms).
WGRAD_STREAM = os.environ.get("SEPT_WGRAD_STREAM", "1") != "0"
# BatchNorm backward: channel sums from the pooled tensors (SEPT_BN_POOLED=0: from every window of the pre-activations)
BN_POOLED_SUMS = os.environ.get("SEPT_BN_POOLED", "1") != "0"
# BatchNorm backward sums of blocks 1 / 2 from the epilogue of the data-gradient conv that produces their dy
# (SEPT_BN_DGRAD_SUMS=0: the separate reduce pass over the pooled tensors)
BN_SUMS_IN_DGRAD = os.environ.get("SEPT_BN_DGRAD_SUMS", "1") != "0"
# block 1's backward pass of a network WITHOUT conv1 weight gradient (the frozen emotion model) from the pooled gradient +
# recorded arg-max positions + the input, no pre-activation-sized tensor (ops.conv1_backward_data_sparse; -1.7 % step
# time).  SEPT_L1_SPARSE=0: BatchNorm backward apply pass + dense data gradient as for the trainable one
L1_SPARSE = os.environ.get("SEPT_L1_SPARSE", "1") != "0"
# block 1 in pool-first form (round 3): conv1's epilogue resolves the 2x2 pooling window BEFORE the BatchNorm (maximum or
# minimum by the sign of gamma) and leaves the statistics partials, so no (B, H, W, 32) tensor exists in either direction.
# SEPT_L1_POOL_FIRST=0: conv1 stores its output and the BatchNorm passes stream it (round 2's form)
L1_POOL_FIRST = os.environ.get("SEPT_L1_POOL_FIRST", "1") != "0"


# ... also for a network whose conv1 is TRAINED (its weight gradient from the pooled gradient, the position bytes and the 26 x 26
# Gram matrix of the input patches: sept_conv1_backward_weight_sparse).  SEPT_L1_POOL_TRAIN=0: only where conv1 is frozen
L1_POOL_TRAINABLE = os.environ.get("SEPT_L1_POOL_TRAIN", "1") != "0"


# the cloak step only needs the input gradient summed over the batch: a pool-first block 1 forms that sum without any
# per-sample data-gradient pass (sept_conv1_backward_data_sum).  SEPT_L1_DX_SUM=0: per-sample gradients, summed by the cloak kernel
L1_DX_SUM = os.environ.get("SEPT_L1_DX_SUM", "1") != "0"


# blocks 2 / 3 of a network whose conv weights need no gradient (the frozen emotion model): the BatchNorm backward apply
# pass inside the tile loader of the data-gradient conv behind it (ops.conv5x5_dgrad_bnapply) -- the (B, H, W, C) gradient of
# the pre-activations is neither written nor read (0.34 GB per step at 224 windows).  SEPT_BN_APPLY_CONV=0: separate pass
BN_APPLY_IN_CONV = os.environ.get("SEPT_BN_APPLY_CONV", "1") != "0"
# ... and, forward, a pool-first block 1's activation pass inside the tile loader of the conv behind it, when that conv needs
# no weight gradient (ops.conv5x5_forward_act): the (B, H/2, W/2, 32) activation is neither written nor read.
# SEPT_BN_ACT_CONV=0: the quarter-size elementwise pass
BN_ACT_IN_CONV = os.environ.get("SEPT_BN_ACT_CONV", "1") != "0"


def batch_sum_pair(dx1, dx2):
    """the two branches' input gradients as equally shaped (rows, n) matrices for sept_cloak_backward: when one of them is
    already summed over the batch (1 row) and the other is not, the other is summed too (one column-sum launch pair)"""
    a, b = dx1.view(dx1.shape[0], -1), (None if dx2 is None else dx2.view(dx2.shape[0], -1))
    if b is None or a.shape[0] == b.shape[0]:
        return a, b
    if a.shape[0] != 1:
        a = ops.colsum(a.contiguous()).view(1, -1)
    if b.shape[0] != 1:
        b = ops.colsum(b.contiguous()).view(1, -1)
    return a, b


def _pool_first_backward_ok(cv, need_grad):
    """the pool-first form has a backward pass without conv1's stored output: always the sparse data gradient; the weight
    gradient of a trainable conv1 through sept_conv1_backward_weight_sparse (L1_POOL_TRAINABLE)"""
    return not need_grad or not cv.weight.requires_grad or L1_POOL_TRAINABLE


# BatchNorm statistics of the 5x5 conv layers from the conv kernel's epilogue (SEPT_CONV_STATS=0: a separate pass)
CONV_FUSED_STATS = os.environ.get("SEPT_CONV_STATS", "1") != "0"
NO_WGRAD_FORK = set()
_WG_STREAMS = {}
_DEFERRED = {"on": False, "pending": []}
# functional.grl_train_step: this many of the trainable branch's 5 x 5 weight gradients (layer 3 first) run inside the
# FROZEN branch's chain (in front of its block 1) instead of inside their own: the trainable branch is the longer one by
# exactly its weight gradients (measured: it finished 240 us after the frozen one), so the two backward chains then end
# within 90 us of each other (-1.2 % step time; 2 or 3 moved: no better).  SEPT_WGRAD_TAIL=0 keeps them at home.
TAIL_WGRADS = int(os.environ.get("SEPT_WGRAD_TAIL", "1"))
_TAIL_WGRADS = {"list": None, "max": 0, "ran": 0, "taken": 0}

# HIP-graph capture and side streams.  On ROCm 7.2 hipStreamEndCapture aborts the PROCESS (core dump, no error
# code) when a forked stream is joined into another FORKED stream inside a capture:
#     origin -> s1 (s1.wait_stream(origin)) -> wg (wg.wait_stream(s1)); s1.wait_stream(wg)      aborts
#     origin -> s1 -> wg; origin.wait_stream(wg); origin.wait_stream(s1)                           works
#     origin -> wg1, origin -> wg2; origin joins both                                             works
# (tools/repro_capture_nested_join.py; gpurun_out/cap_*.log of round 1).  So under capture a side stream may
# only be used when its join lands on the capture's ORIGIN stream.  The trainers' capture() methods publish that
# stream through capture_origin(); everything here that forks checks fork_allowed() and otherwise runs in line --
# same kernels, same results, one queue -- instead of recording a topology that kills the process.
_CAPTURE = {"origin": None}


class capture_origin:
    """with capture_origin(): ... -- marks the current stream as the origin of a HIP-graph capture."""

    def __enter__(self):
        self.prev = _CAPTURE["origin"]
        _CAPTURE["origin"] = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        _CAPTURE["origin"] = self.prev
        return False


def fork_allowed(device, join_on_current=True):
    """May the caller fork a side stream off the CURRENT stream (and join it back there)?  Always outside a
    capture; inside one only when the current stream is the published capture origin."""
    if device.type != "cuda" or not torch.cuda.is_current_stream_capturing():
        return True
    return _CAPTURE["origin"] is not None and torch.cuda.current_stream(device).cuda_stream == _CAPTURE["origin"]


def backward(loss):
    """loss.backward() for the trainers: branch networks may leave small weight gradients on side streams;
    they are joined into the current stream here, before the caller touches any .grad.  Under a capture
    the deferred join is only recorded when this stream is the capture origin (see above)."""
    _DEFERRED["on"] = WGRAD_STREAM and fork_allowed(loss.device)
    try:
        loss.backward()
    finally:
        _DEFERRED["on"] = False
        cur = torch.cuda.current_stream()
        for wg, _keep in _DEFERRED["pending"]:
            cur.wait_stream(wg)
        _DEFERRED["pending"].clear()


class _SideQueue:
    """The weight-gradient side stream of ONE network's backward pass (see the comment above)."""

    def __init__(self, device, enabled):
        self.dev, self.wg, self.nested, self.keep = device, None, False, []
        if not (enabled and WGRAD_STREAM and device.type == "cuda"):
            return
        cur = torch.cuda.current_stream(device).cuda_stream
        self.nested = cur in NO_WGRAD_FORK
        if self.nested and not _DEFERRED["on"]:
            return
        if not self.nested and not fork_allowed(device):
            return   # capturing on a non-origin stream: the join below would be a nested one -- stay in line
        key = (device.index, cur)
        if key not in _WG_STREAMS:
            _WG_STREAMS[key] = torch.cuda.Stream(device=device)
        self.wg = _WG_STREAMS[key]

    def _run(self, fn, operands):
        self.wg.wait_stream(torch.cuda.current_stream(self.dev))   # after everything enqueued so far
        self.keep.extend(operands)                                  # operands stay referenced until the join
        with torch.cuda.stream(self.wg):
            return fn()

    def small(self, fn, *operands):
        """latency-bound weight-gradient launches: on the side stream whenever there is one"""
        return fn() if self.wg is None else self._run(fn, operands)

    def big(self, fn, *operands):
        """MFMA-heavy weight gradients: on the side stream of a top-level network only"""
        return fn() if (self.wg is None or self.nested) else self._run(fn, operands)

    def finish(self):
        if self.wg is None:
            return
        if self.nested:   # functional.backward() joins
            _DEFERRED["pending"].append((self.wg, self.keep))
        else:             # the caller (autograd) sees every gradient on this network's stream
            torch.cuda.current_stream(self.dev).wait_stream(self.wg)
            self.keep.clear()


def _run_tail_wgrads(device):
    """Run the weight gradients the OTHER branch handed over (trunk_backward with _TAIL_WGRADS set), on the current stream:
    each waits for the event that marks its operands ready."""
    tail = _TAIL_WGRADS["list"]
    if not tail:
        return
    here = torch.cuda.current_stream(device)
    capturing = torch.cuda.is_current_stream_capturing()
    while tail and tail[0][0] is not None:
        ev, fn, operands = tail.pop(0)
        here.wait_event(ev)
        if not capturing:
            for t in operands:
                t.record_stream(here)
        fn()
        _TAIL_WGRADS["ran"] += 1


def _trunk_backward_rnn_conv(S, P, dout, grads, put, need_wgrad, need_dx, sq, gout, sum_dx=False):
    """The recurrent layers and the conv stack of trunk_backward, from the gradient of the last recurrent output."""
    B, T = S.B, S.T
    Hh = S.Hh
    H2, G = 2 * Hh, (4 if S.lstm else 3) * Hh
    r = P.rnn
    dseq = None
    for layer in (1, 0):
        Gs = S.gru[layer]
        sfx = f"_l{layer}"
        if S.lstm:   # gi and W_hh h enter the gates as a sum: one gradient serves both
            dgi, hprev = ops.lstm_backward(dout.contiguous(), Gs.out, Gs.gates, Gs.cells, Gs.whf, Gs.whr)
            dgh = dgi
        else:        # layer 0: the inter-layer dropout's backward (dout * mask) happens as the kernel fetches dout
            dgi, dgh, hprev = ops.gru_backward(dout.contiguous(), Gs.out, Gs.gates, Gs.whf, Gs.whr,
                                               dout_mask=Gs.mask if layer == 0 else None)
        dgi2, dgh2, hp2 = dgi.view(B * T, 2 * G), dgh.view(B * T, 2 * G), hprev.view(B * T, H2)
        K = Gs.inp.shape[1]
        if need_wgrad:
            def rnn_wgrads(layer=layer, sfx=sfx, Gs=Gs, dgi2=dgi2, dgh2=dgh2, hp2=hp2):
                # both directions in one product / one column sum each; forward and reverse parameters sit next to
                # each other in the trainer's flat gradient buffer, so the (2G, ...) results are written in place
                pair = (lambda n: grad_out_pair(getattr(r, n + sfx), getattr(r, n + sfx + "_reverse"))) if need_wgrad \
                    else (lambda n: None)
                o_w, o_bi, o_bh = (pair("weight_ih") if layer == 1 else None), pair("bias_ih"), pair("bias_hh")
                # bias gradients = column sums of the gate gradients: formed by the weight-gradient products themselves
                # (ops.linear_backward_weight(colsum_out=...)), no launches of their own
                dev_ = dgi2.device
                dbih = o_bi if o_bi is not None else torch.empty(2 * G, dtype=torch.float32, device=dev_)
                dwcat = ops.linear_backward_weight(dgi2, Gs.inp, out=o_w, colsum_out=dbih)      # (2G, K)
                dbhh = dbih if S.lstm else (o_bh if o_bh is not None else torch.empty(2 * G, dtype=torch.float32, device=dev_))
                if S.lstm and o_bh is not None:
                    dbhh = ops.copy_into(o_bh, dbih)
                for d, tag in ((0, ""), (1, "_reverse")):
                    gh = dgh2[:, d * G:(d + 1) * G]
                    dwih = dwcat[d * G:(d + 1) * G]
                    wih, whh = getattr(r, "weight_ih" + sfx + tag), getattr(r, "weight_hh" + sfx + tag)
                    bih, bhh = getattr(r, "bias_ih" + sfx + tag), getattr(r, "bias_hh" + sfx + tag)
                    if layer == 0:
                        dwih = ops.permute_cols(dwih, S.C, S.Wd, inverse=True, out=gout(wih))
                    elif o_w is None:
                        dwih = _into(gout(wih), dwih)
                    put(wih, dwih)
                    put(whh, ops.linear_backward_weight(gh, hp2[:, d * Hh:(d + 1) * Hh], out=gout(whh),
                                                        colsum_out=None if S.lstm else dbhh[d * G:(d + 1) * G]))
                    put(bih, dbih[d * G:(d + 1) * G] if o_bi is not None else _into(gout(bih), dbih[d * G:(d + 1) * G]))
                    put(bhh, dbhh[d * G:(d + 1) * G] if o_bh is not None else _into(gout(bhh), dbhh[d * G:(d + 1) * G]))
            sq.small(rnn_wgrads, dgi2, dgh2, hp2, Gs.inp)
        # gradient wrt the layer input: dgi [W_if; W_ir]  (one product, reduction length 2G)
        odt = torch.float32 if layer == 1 else torch.bfloat16
        din = ops.linear_nt_split(dgi2, Gs.wcatT, None, out_dtype=odt)
        if layer == 1:
            dout = din.view(B, T, H2)
            if S.gru[0].mask is not None and S.lstm:
                dout = ops.mul(dout, S.gru[0].mask)
        else:
            dseq = din
    # ---- conv stack ----
    dact = dseq.view(B, T, S.Wd, S.C)
    dx = None
    presums = None   # (partials, count) of the current block's BatchNorm backward sums when the producer of dact formed them
    for li in range(len(S.blocks) - 1, -1, -1):
        blk, cv, bn = S.blocks[li], P.convs[li], P.bns[li]
        if not blk.bn_train:
            raise SeptError("backward through an eval-mode BatchNorm is not implemented on the HIP path")
        if li == 0 and not need_wgrad:
            _run_tail_wgrads(dact.device)   # the other branch's deferred weight gradients, in front of this branch's block 1
        want_bn = need_wgrad and bn.weight.requires_grad
        if li == 0 and blk.ext is not None:
            # pool-first block 1: channel sums from (dy, ext, idx) -- formed by the data-gradient conv above when it could --
            # then the data gradient from the pooled gradient, the position bytes and the input
            sums, dgamma, dbeta = ops.bn_backward_sums_ext(dact, blk.ext, blk.mean, blk.invstd, bn.weight, bn.bias, blk.drop,
                                                           presums, need_param_grads=want_bn,
                                                           out_gamma=gout(bn.weight) if want_bn else None,
                                                           out_beta=gout(bn.bias) if want_bn else None)
            if want_bn:
                put(bn.weight, dgamma)
                put(bn.bias, dbeta)
            if need_wgrad and cv.weight.requires_grad:
                dw, _ = sq.big(lambda dact=dact, sums=sums: ops.conv1_backward_weight_from_sums(
                    S.x, dact, blk.idx, sums, blk.mean, blk.invstd, bn.weight, blk.drop, cv.weight, cv.bias, need_bias=False,
                    out_w=gout(cv.weight)), dact, sums, S.x, blk.idx)
                put(cv.weight, dw)
                if cv.bias is not None:
                    put(cv.bias, zero_bias_grad(cv.bias))
            if need_dx and sum_dx and L1_DX_SUM:
                dx = ops.conv1_backward_data_sum(S.x, dact, blk.idx, sums, blk.mean, blk.invstd, bn.weight, blk.drop,
                                                 cv.weight, cv.bias)
            elif need_dx:
                dx = ops.conv1_backward_data_from_sums(S.x, dact, blk.idx, sums, blk.mean, blk.invstd, bn.weight, blk.drop,
                                                       cv.weight, cv.bias, prep=_conv1_operand(cv))
            continue
        if (li == 0 and getattr(blk, "idx", None) is not None and need_dx and not blk.sync
                and not (need_wgrad and cv.weight.requires_grad)):
            # block 1 of a network without conv1 weight gradient (the frozen emotion model): no pre-activation-sized tensor
            # is read or written -- sparse part on the MFMA data-gradient kernel, dense part as a linear map of the input
            dx, dgamma, dbeta = ops.conv1_backward_data_sparse(
                S.x, blk.pre, dact, blk.idx, blk.mean, blk.invstd, bn.weight, bn.bias, blk.drop, cv.weight, cv.bias,
                presums=presums, need_param_grads=want_bn, out_gamma=gout(bn.weight) if want_bn else None,
                out_beta=gout(bn.bias) if want_bn else None, prep=_conv1_operand(cv), y=blk.out if BN_POOLED_SUMS else None)
            if want_bn:
                put(bn.weight, dgamma)
                put(bn.bias, dbeta)
            continue
        if (li >= 1 and BN_APPLY_IN_CONV and not (need_wgrad and cv.weight.requires_grad) and blk.pool == 2 and not blk.sync
                and blk.pre.shape[1] % 2 == 0 and blk.pre.shape[2] % 2 == 0):
            prev, pbn, cin_ = S.blocks[li - 1], P.bns[li - 1], cv.weight.shape[1]
            ep = None   # what the conv's epilogue leaves for the block in front (as the unfused calls below)
            if prev.ext is not None:
                if cin_ <= 32 and tuple(prev.ext.shape) == tuple(blk.pre.shape[:3]) + (cin_,):
                    ep = ("ext", prev.ext, prev.mean, prev.invstd, pbn.weight, pbn.bias, prev.drop)
            elif (BN_SUMS_IN_DGRAD and BN_POOLED_SUMS and prev.bn_train and not prev.sync
                    and prev.pool == 2 and tuple(prev.out.shape) == tuple(blk.pre.shape[:3]) + (cin_,)):
                ep = ("pool", prev.out, pbn.weight, pbn.bias, prev.drop)
            if ep is not None and not ops.conv5x5_bnapply_supported(blk.pre, cin_, True):
                ep = None
            if ep is not None or ops.conv5x5_bnapply_supported(blk.pre, cin_, False):
                kw = dict(need_param_grads=want_bn, out_gamma=gout(bn.weight) if want_bn else None,
                          out_beta=gout(bn.bias) if want_bn else None)
                if presums is not None:
                    sums, dgamma, dbeta = ops.bn_backward_sums_presummed(dact, blk.pre, blk.mean, blk.invstd, bn.weight, bn.bias,
                                                                         blk.drop, presums, blk.pool, **kw)
                else:
                    sums, dgamma, dbeta = ops.bn_backward_sums(dact, blk.pre, blk.mean, blk.invstd, bn.weight, bn.bias, blk.drop,
                                                               blk.pool, y=blk.out if BN_POOLED_SUMS else None, **kw)
                if want_bn:
                    put(bn.weight, dgamma)
                    put(bn.bias, dbeta)
                wtd = _cached("convdgrad", cv.weight, lambda: ops.conv5x5_prep_weights(cv.weight, 1))
                dact, presums = ops.conv5x5_dgrad_bnapply(blk.pre, dact, sums, blk.mean, blk.invstd, bn.weight, bn.bias,
                                                          blk.drop, wtd, ep)
                continue
        if presums is not None:
            dpre, dgamma, dbeta = ops.bn_relu_pool_backward_presummed(dact, blk.pre, blk.mean, blk.invstd, bn.weight,
                                                                      bn.bias, blk.drop, presums, blk.pool,
                                                                      need_param_grads=want_bn,
                                                                      out_gamma=gout(bn.weight) if want_bn else None,
                                                                      out_beta=gout(bn.bias) if want_bn else None)
        else:
            dpre, dgamma, dbeta = ops.bn_relu_pool_backward(dact, blk.pre, blk.mean, blk.invstd, bn.weight, bn.bias,
                                                            blk.drop, blk.pool, need_param_grads=want_bn,
                                                            sync_group=_SYNC_BN["group"], sync=blk.sync,
                                                            y=blk.out if BN_POOLED_SUMS else None,
                                                            out_gamma=gout(bn.weight) if want_bn else None,
                                                            out_beta=gout(bn.bias) if want_bn else None)
        if want_bn:
            put(bn.weight, dgamma)
            put(bn.bias, dbeta)
        if li == 0:
            if need_wgrad and cv.weight.requires_grad:
                dw, _ = sq.big(lambda dpre=dpre: ops.conv1_backward_weight(S.x, dpre, need_bias=False, out_w=gout(cv.weight)),
                               dpre, S.x)
                put(cv.weight, dw)
                if cv.bias is not None:     # in front of a training-mode BatchNorm: exactly zero (see zero_bias_grad)
                    put(cv.bias, zero_bias_grad(cv.bias))
            if need_dx:
                dx = ops.conv1_backward_data(dpre, cv.weight, prep=_conv1_operand(cv))
        else:
            if need_wgrad and cv.weight.requires_grad:
                tail = _TAIL_WGRADS["list"]
                if tail is not None and _TAIL_WGRADS["taken"] < _TAIL_WGRADS["max"]:
                    _TAIL_WGRADS["taken"] += 1
                    # handed to the caller, which runs it at the END of the OTHER branch's chain (grl_train_step): the
                    # trainable branch is the longer one by exactly its weight gradients
                    dw = gout(cv.weight)
                    dw = dw if dw is not None else torch.empty_like(cv.weight)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dpre.device))
                    tail.append((ev, lambda blk=blk, dpre=dpre, dw=dw: ops.conv5x5_backward_weight(blk.inp, dpre, out=dw),
                                 (dpre, blk.inp)))
                    put(cv.weight, dw)
                else:
                    put(cv.weight, sq.big(lambda blk=blk, dpre=dpre: ops.conv5x5_backward_weight(blk.inp, dpre,
                                                                                                 out=gout(cv.weight)),
                                          dpre, blk.inp))
                if cv.bias is not None:
                    put(cv.bias, zero_bias_grad(cv.bias))
            wtd = _cached("convdgrad", cv.weight, lambda: ops.conv5x5_prep_weights(cv.weight, 1))
            prev = S.blocks[li - 1]
            if prev.ext is not None:
                # the producer masks the gradient (zero where block 1's ReLU is inactive) and leaves the sums; without such
                # a kernel form the plain conv runs and ops.bn_backward_sums_ext masks + sums in the next iteration
                pbn = P.bns[li - 1]
                dact, presums = ops.conv5x5_dgrad_bnsums_ext(dpre, wtd, prev.ext, prev.mean, prev.invstd, pbn.weight, pbn.bias,
                                                             prev.drop)
                if dact is None:
                    dact, presums = ops.conv5x5(dpre, wtd), None
            elif BN_SUMS_IN_DGRAD and BN_POOLED_SUMS and prev.bn_train and not prev.sync and prev.pool == 2:
                # the data-gradient conv's epilogue also forms the backward sums of the BatchNorm in front (from its
                # output tile and that block's pooled activation): that block's reduce pass over y / dy disappears
                pbn = P.bns[li - 1]
                dact, presums = ops.conv5x5_dgrad_bnsums(dpre, wtd, prev.out, pbn.weight, pbn.bias, prev.drop)
            else:
                dact, presums = ops.conv5x5(dpre, wtd), None
    sq.finish()
    return dx, grads


# ---------------------------------------------------------------------------------------------
# autograd glue
# ---------------------------------------------------------------------------------------------
def _param_list(P):
    ps = []
    for cv, bn in zip(P.convs, P.bns):
        ps += [cv.weight, cv.bias, bn.weight, bn.bias]
    ps += list(P.rnn.parameters())
    ps += [P.dense1.weight, P.dense1.bias]
    for h in P.heads:
        ps += [h.weight, h.bias]
    if P.att:
        ps += [p for p in (P.att1.weight, P.att1.bias, P.att2.weight, P.att2.bias) if p is not None]
    return ps


class TrunkFn(torch.autograd.Function):
    """logits = trunk(x; params).  forward(ctx, x, P, pooling, injected, gfeat, *params)."""

    @staticmethod
    def forward(ctx, x, P, pooling, injected, gfeat, *params):
        need = any(ctx.needs_input_grad)
        logits, S = trunk_forward(x.detach().contiguous().view(x.shape[0], x.shape[-2], x.shape[-1]), P, pooling,
                                  need_grad=need, injected=injected, gfeat=gfeat)
        ctx.S, ctx.P, ctx.params, ctx.xshape = S, P, params, x.shape
        ctx.need_dx = x.requires_grad
        ctx.need_w = any(p.requires_grad for p in params)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        dx, grads = trunk_backward(ctx.S, ctx.P, dlogits, need_wgrad=ctx.need_w, need_dx=ctx.need_dx)
        ctx.S = None
        gp = tuple(grads.get(p) if p.requires_grad else None for p in ctx.params)
        return (dx.view(ctx.xshape) if dx is not None else None, None, None, None, None) + gp


def run_trunk(model, x, head, pooling="mean", injected=None, gfeat=None, att="model"):
    """head 'multitask' returns the (B, 4 + 2) logits of both prediction layers side by side."""
    P = trunk_params(model, head, att)
    return TrunkFn.apply(x, P, pooling, injected, gfeat, *_param_list(P))


class GradientReversalFunction(torch.autograd.Function):
    """reversal_gradient.py:5-23: identity forward, -lambda * grad backward (HIP scale kernel)."""

    @staticmethod
    def forward(ctx, x, lambda_):
        ctx.lambda_ = lambda_
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grads):
        require_cuda(grads)
        return ops.scale(grads.contiguous().float(), -float(ctx.lambda_)), None


class CloakFn(torch.autograd.Function):
    """cloak_noise.forward (cloak_models.py:52-58) with an explicit epsilon."""

    @staticmethod
    def forward(ctx, x, locs, rhos, eps, mask, min_scale, max_scale):
        xn = ops.cloak_forward(x.detach().float().contiguous(), locs.detach(), rhos.detach(), eps, mask,
                               min_scale, max_scale)
        ctx.save_for_backward(rhos.detach(), eps, mask if mask is not None else torch.empty(0))
        ctx.has_mask = mask is not None
        ctx.per_row = eps.numel() != rhos.numel()
        ctx.cfg = (float(min_scale), float(max_scale))
        ctx.need = (locs.requires_grad, rhos.requires_grad)
        return xn

    @staticmethod
    def backward(ctx, dxn):
        rhos, eps, mask = ctx.saved_tensors
        if ctx.per_row:
            raise SeptError("per-window epsilon is an inference mode (test() loops); training draws one epsilon per step")
        dlocs, drhos = ops.cloak_backward(dxn.contiguous(), None, 0.0, rhos, eps, mask if ctx.has_mask else None,
                                          ctx.cfg[0], ctx.cfg[1], need_locs=ctx.need[0], need_rhos=ctx.need[1])
        return None, dlocs, drhos, None, None, None, None


# run the emotion and the gender branch of the GRL step on two HIP streams (SEPT_CONCURRENT=0 disables)
CONCURRENT_BRANCHES = os.environ.get("SEPT_CONCURRENT", "1") != "0"
_BRANCH_STREAMS = {}


def branch_streams(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _BRANCH_STREAMS:
        _BRANCH_STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        NO_WGRAD_FORK.update(st.cuda_stream for st in _BRANCH_STREAMS[key])   # see WGRAD_STREAM above
    return _BRANCH_STREAMS[key]


class GrlPairFn(torch.autograd.Function):
    """The whole body of two_d_cnn_lstm_syn_with_grl.forward (cloak_models.py:157-226) as ONE autograd node:
        xn = cloak(x);  preds1 = emotion_trunk(xn);  preds2 = gender_trunk(GradientReversal(xn))
    Forward and backward run the two trunks on two HIP streams (they only share xn), joined on the caller's
    stream inside this node -- so under a HIP-graph capture every join lands on the capture origin.  The
    backward hands BOTH branches' input gradients to one cloak kernel, sept_cloak_backward(dxa, dxb, -lambda):
    no gradient-reversal scale pass, no autograd accumulation add (three full-tensor passes in round 1).
    forward(ctx, x, locs, rhos, eps, mask, cfg, P1, P2, pooling, gfeat, n1, *params): cfg = (min_scale, max_scale,
    grl_lambda); params = the n1 tensors of the emotion trunk, then the gender trunk's."""

    @staticmethod
    def forward(ctx, x, locs, rhos, eps, mask, cfg, P1, P2, pooling, gfeat, n1, *params):
        smin, smax, lam = cfg
        shape = x.shape
        x2 = x.detach().float().contiguous().view(shape[0], -1)
        xn = ops.cloak_forward(x2, locs.detach(), rhos.detach(), eps, mask, smin, smax)
        xw = xn.view(shape[0], shape[-2], shape[-1])
        need = any(ctx.needs_input_grad)
        dev = x.device
        two = CONCURRENT_BRANCHES and fork_allowed(dev)
        if two:
            cur = torch.cuda.current_stream(dev)
            s1, s2 = branch_streams(dev)
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            capturing = torch.cuda.is_current_stream_capturing()   # graph-private memory needs no stream records
            if not capturing:
                xn.record_stream(s1)
                xn.record_stream(s2)
            with torch.cuda.stream(s1):
                l1, S1 = trunk_forward(xw, P1, pooling, need_grad=need, gfeat=gfeat, rng_site=SITE_EMOTION)
            with torch.cuda.stream(s2):
                l2, S2 = trunk_forward(xw, P2, pooling, need_grad=need, gfeat=gfeat, rng_site=SITE_GENDER)
            cur.wait_stream(s1)
            cur.wait_stream(s2)
            if not capturing:
                l1.record_stream(cur)
                l2.record_stream(cur)
        else:
            l1, S1 = trunk_forward(xw, P1, pooling, need_grad=need, gfeat=gfeat, rng_site=SITE_EMOTION)
            l2, S2 = trunk_forward(xw, P2, pooling, need_grad=need, gfeat=gfeat, rng_site=SITE_GENDER)
        ctx.S, ctx.P, ctx.two = (S1, S2), (P1, P2), two
        ctx.params, ctx.n1, ctx.cfg = params, n1, cfg
        ctx.cloak = (locs, rhos, eps, mask)
        ctx.per_row = eps.numel() != rhos.numel()
        ctx.need_cloak = (locs.requires_grad, rhos.requires_grad)
        ctx.need_w = (any(p.requires_grad for p in params[:n1]), any(p.requires_grad for p in params[n1:]))
        noisy = xn.view(shape)
        ctx.mark_non_differentiable(noisy)     # the reference returns input.detach() (:162)
        return l1, l2, noisy

    @staticmethod
    def backward(ctx, d1, d2, _dnoisy):
        (S1, S2), (P1, P2) = ctx.S, ctx.P
        locs, rhos, eps, mask = ctx.cloak
        smin, smax, lam = ctx.cfg
        need_dx = any(ctx.need_cloak)
        if need_dx and ctx.per_row:   # as CloakFn.backward: the cloak backward kernel reads ONE epsilon for the batch
            raise SeptError("per-window epsilon is an inference mode (test() loops); training draws one epsilon per step")
        dev = d1.device
        two = ctx.two and fork_allowed(dev)
        prev = _DEFERRED["on"]
        if two:
            cur = torch.cuda.current_stream(dev)
            s1, s2 = branch_streams(dev)
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            _DEFERRED["on"] = WGRAD_STREAM     # the branches' small weight gradients fork; joined HERE, on `cur`
            try:
                with torch.cuda.stream(s2):    # the gender branch first: its recurrent chain is the critical path
                    dx2, g2 = trunk_backward(S2, P2, d2, need_wgrad=ctx.need_w[1], need_dx=need_dx, sum_dx=True)
                with torch.cuda.stream(s1):
                    dx1, g1 = trunk_backward(S1, P1, d1, need_wgrad=ctx.need_w[0], need_dx=need_dx, sum_dx=True)
            finally:
                _DEFERRED["on"] = prev
            cur.wait_stream(s1)
            cur.wait_stream(s2)
            for wg, _keep in _DEFERRED["pending"]:
                cur.wait_stream(wg)
            _DEFERRED["pending"].clear()
            if not torch.cuda.is_current_stream_capturing():   # allocated on s1 / s2, consumed on cur
                for t in (dx1, dx2):
                    if t is not None:
                        t.record_stream(cur)
        else:
            _DEFERRED["on"] = False
            try:
                dx2, g2 = trunk_backward(S2, P2, d2, need_wgrad=ctx.need_w[1], need_dx=need_dx, sum_dx=True)
                dx1, g1 = trunk_backward(S1, P1, d1, need_wgrad=ctx.need_w[0], need_dx=need_dx, sum_dx=True)
            finally:
                _DEFERRED["on"] = prev
        ctx.S = None
        dlocs = drhos = None
        if need_dx:
            da, db_ = batch_sum_pair(dx1, dx2)
            dlocs, drhos = ops.cloak_backward(da, db_, -float(lam), rhos.detach(), eps, mask,
                                              smin, smax, need_locs=ctx.need_cloak[0], need_rhos=ctx.need_cloak[1],
                                              out_locs=grad_out(locs) if ctx.need_cloak[0] else None,
                                              out_rhos=None)   # rhos may get a second term (scale loss): autograd adds
        grads = dict(g1)
        grads.update(g2)
        gp = tuple(grads.get(p) if p.requires_grad else None for p in ctx.params)
        return (None, dlocs, drhos, None, None, None, None, None, None, None, None) + gp


def grl_train_step(model, x, labels_emo, labels_gen, weights, gender_lambda, scale_lamda, use_scale_term=True, mask=None,
                   pooling="mean", global_feature=None, before_cloak=None, injected=None, at_join=None):
    """One forward + loss + backward of two_d_cnn_lstm_syn_with_grl under the loss of train()
    (training_cloak_with_grl.py:122-160), scheduled by hand instead of through the autograd tape:

        cur:  [x = before_cloak()]   cloak -> xn ........................................ cloak backward(dx1, dx2, -lambda)
        s1:                 emotion trunk(xn) -> CE -> d logits -> emotion data gradient dx1 ....^
        s2:   rng, epsilon  gender trunk(xn)  -> CE -> d logits -> gender data + weight gradients dx2 ^

    Each branch runs forward, its cross-entropy and its backward as ONE chain on its own stream: the emotion branch
    (frozen: no operand preparation, no weight gradients) does not wait for the gender forward before it starts its
    backward, so the launch-bound recurrent / head sections of one branch overlap the convolutions of the other, and the
    loss section is one kernel per branch (no root-gradient fill, no upstream-gradient scale).  Same kernels, same
    gradient slots as the autograd path (GrlPairFn + GrlStepLossFn), which computes exactly this.  `x` is (B, 1, H, W)
    or None with `before_cloak` a callable that produces it on the current stream (the feature stage: its launches then
    overlap the step's random-number kernels).  Returns (loss, logits_emotion, logits_gender); the gradients are in
    .grad of the trainable parameters (views of their flat slots where the trainer packed them).  `injected` (default: the
    wrapper's `injected_masks` test hook): explicit dropout masks (emotion network's, gender network's), see trunk_forward.
    `at_join`: called on the current stream where both backward chains (and every side stream) have been joined, in front of
    the cloak backward kernel -- from here on every weight gradient of the two networks is final in its slot; only dL/dlocs
    and dL/drhos are still to come.  The data-parallel trainer uses it to start the all-reduce of the first gradient bucket
    (and, under capture, to end the first graph segment there)."""
    noise, emo, gen = model.intermed, model.original_model, model.gender_model
    injected = injected if injected is not None else getattr(model, "injected_masks", None)
    inj1, inj2 = injected if injected is not None else (None, None)
    att = emo.att
    pool = "flatten" if pooling is None else "mean"
    locs, rhos = noise.locs, noise.rhos
    dev = rhos.device
    cur = torch.cuda.current_stream(dev)
    two = CONCURRENT_BRANCHES and fork_allowed(dev)
    capturing = torch.cuda.is_current_stream_capturing()
    s1, s2 = branch_streams(dev)
    with torch.no_grad():
        ops.stamp("step start", dev)
        # ---- random numbers of the step on s2 while the caller's feature stage runs on cur: the cloak's epsilon and -- so
        # that neither branch's chain opens with a mask launch -- the dropout masks of both networks (their own Philox call
        # sites: the same values trunk_forward would draw itself)
        P1, P2 = trunk_params(emo, 'emotion', att), trunk_params(gen, 'gender', att)
        pre_masks = {}

        def draws(xs=None):
            ops.begin_step(dev)
            e = noise._epsilon(1)
            if xs is not None:
                for P, site, inj in ((P1, SITE_EMOTION, inj1), (P2, SITE_GENDER, inj2)):
                    if P.training:
                        pre_masks[id(P)] = step_masks(P, xs[0], xs[-2], dev, inj or getattr(P, "injected", None), site)
            return e
        if two and before_cloak is not None:
            s2.wait_stream(cur)             # recorded BEFORE the feature stage is enqueued: s2 runs beside it
            x = before_cloak()
            with torch.cuda.stream(s2):
                eps = draws(x.shape)
            cur.wait_stream(s2)
            if not capturing:
                eps.record_stream(cur)
                for md in pre_masks.values():
                    for t in md.values():
                        t.record_stream(s1), t.record_stream(s2)
        else:
            eps = draws()
            if before_cloak is not None:
                x = before_cloak()
        shape = x.shape
        B = shape[0]
        m = None if mask is None else mask.to(dev, torch.float32).contiguous()
        smin, smax, lam = float(noise.min_scale), float(noise.max_scale), float(gen.conv[0].lambda_)
        if isinstance(x, ops.LazyWindows):   # the feature stage's windows are formed inside the cloak kernel
            xn = ops.window_norm_cloak(x, locs.detach(), rhos.detach(), eps, m, smin, smax)
        else:
            xn = ops.cloak_forward(x.detach().float().contiguous().view(B, -1), locs.detach(), rhos.detach(), eps, m, smin, smax)
        xw = xn.view(B, shape[-2], shape[-1])
        ops.stamp("cloak forward done")
        need_dx = locs.requires_grad or rhos.requires_grad
        need_w2 = any(p.requires_grad for p in _param_list(P2))
        need_w1 = any(p.requires_grad for p in _param_list(P1))
        scale_mean = None
        if use_scale_term and float(scale_lamda) != 0.0:
            _, scale_mean = ops.cloak_scales(rhos.detach(), smin, smax, want_scales=False, want_mean=True)
        loss_a = torch.empty((), dtype=torch.float32, device=dev)
        loss_b = torch.empty((), dtype=torch.float32, device=dev)

        def fwd(P):
            tag = "emotion" if P is P1 else "gender"
            ops.stamp(tag + " forward starts")
            r = trunk_forward(xw, P, pool, need_grad=True, gfeat=global_feature, masks=pre_masks.get(id(P)),
                              rng_site=SITE_EMOTION if P is P1 else SITE_GENDER, injected=inj1 if P is P1 else inj2)
            ops.stamp(tag + " forward done")
            return r

        # The loss VALUE is needed by nobody inside the step: where the head's backward kernel can form the cross-entropy
        # gradient itself, the loss kernels of both networks leave the chains' heads and run at the END of the frozen
        # branch's chain (the shorter one), behind the join that follows the forward passes.
        loss_jobs = []
        loss_sum = None

        def bwd(P, logits, S, labels, coef, loss_slot, need_w, with_scale):
            tag = "emotion" if P is P1 else "gender"
            ops.stamp(tag + " backward starts")

            def loss_value(want_grad):
                d_ = ops.cross_entropy(logits, labels, weights, coef / B, loss_slot, want_grad=want_grad)
                if with_scale and scale_mean is not None:
                    ops.loss_sub_log(loss_slot, scale_mean, float(scale_lamda))
                return d_
            if two and (need_w or need_dx) and head_ce_fusable(S, P):
                loss_jobs.append((lambda: loss_value(False), logits))
                r = trunk_backward(S, P, None, need_wgrad=need_w, need_dx=need_dx, sum_dx=True,
                                   ce=(logits, labels, weights, coef / B))
            else:
                d = loss_value(True)
                r = trunk_backward(S, P, d, need_wgrad=need_w, need_dx=need_dx, sum_dx=True) if (need_w or need_dx) else (None, {})
            ops.stamp(tag + " backward done")
            return r

        # (network, labels, loss coefficient, loss slot, weight gradients?, carries the scale term?); the gender chain is
        # enqueued first (enqueueing the emotion branch first measured +2 %) -- the dropout draws of a network use its own
        # Philox call site (SITE_EMOTION / SITE_GENDER), so the masks do not depend on the enqueue order, nor on
        # hand-scheduled vs autograd, capture vs eager
        emo_args = (P1, labels_emo, 1.0, loss_a, need_w1, True)
        gen_args = (P2, labels_gen, float(gender_lambda), loss_b, need_w2, False)
        order = ((s2, gen_args), (s1, emo_args))
        res = {}
        prev = _DEFERRED["on"]
        if two:
            def fork():
                s1.wait_stream(cur)
                s2.wait_stream(cur)

            def join():
                cur.wait_stream(s1)
                cur.wait_stream(s2)

            fork()
            if not capturing:
                xn.record_stream(s1)
                xn.record_stream(s2)
            _DEFERRED["on"] = WGRAD_STREAM     # the gender branch's small weight gradients fork; joined below, on `cur`
            try:
                # the branches meet after their forward passes, as on the autograd tape (each branch as ONE chain forward ->
                # loss -> backward measured 4 % slower: the graph executor then puts the two long chains on one queue)
                saved = {}
                for st, (P, *_r) in order:
                    with torch.cuda.stream(st):
                        saved[st] = fwd(P)
                join()
                ops.stamp("forward joined")
                fork()
                # the trainable branch first (it hands over weight gradients), the frozen branch picks them up in
                # front of its block 1; whatever is left runs behind the frozen branch
                _TAIL_WGRADS.update(list=[] if TAIL_WGRADS > 0 else None, max=TAIL_WGRADS, ran=0, taken=0)
                try:
                    for st, (P, lab, coef, slot, nw, sc) in order:
                        with torch.cuda.stream(st):
                            logits, S = saved[st]
                            res[st] = (logits,) + bwd(P, logits, S, lab, coef, slot, nw, sc)
                    with torch.cuda.stream(s1):
                        _run_tail_wgrads(dev)
                        for job, lg in loss_jobs:   # both forward passes are behind s1 since the join above
                            if not capturing:
                                lg.record_stream(s1)
                            job()
                        if len(loss_jobs) == 2:     # both loss values were formed here: their sum too, off the step's tail
                            loss_sum = ops.add(loss_a, loss_b)
                finally:
                    _TAIL_WGRADS["list"] = None
                saved.clear()
            finally:
                _DEFERRED["on"] = prev
            join()
            for wg, _keep in _DEFERRED["pending"]:
                cur.wait_stream(wg)
            _DEFERRED["pending"].clear()
            (l1, dx1, g1), (l2, dx2, g2) = res[s1], res[s2]
            if not capturing:
                for t in (l1, l2, dx1, dx2, loss_a, loss_b, loss_sum):
                    if t is not None:
                        t.record_stream(cur)
        else:
            _DEFERRED["on"] = False
            try:
                for st, (P, lab, coef, slot, nw, sc) in order:
                    logits, S = fwd(P)
                    res[st] = (logits,) + bwd(P, logits, S, lab, coef, slot, nw, sc)
            finally:
                _DEFERRED["on"] = prev
            (l1, dx1, g1), (l2, dx2, g2) = res[s1], res[s2]
        if at_join is not None:
            at_join()
        if need_dx:
            da, db_ = batch_sum_pair(dx1, dx2)
            dlocs, drhos = ops.cloak_backward(da, db_, -lam, rhos.detach(), eps, m, smin, smax,
                                              scale_lambda=float(scale_lamda) if scale_mean is not None else 0.0,
                                              scale_mean=scale_mean, need_locs=locs.requires_grad,
                                              need_rhos=rhos.requires_grad,
                                              out_locs=grad_out(locs) if locs.requires_grad else None,
                                              out_rhos=grad_out(rhos) if rhos.requires_grad else None)
            if locs.requires_grad:
                locs.grad = dlocs
            if rhos.requires_grad:
                rhos.grad = drhos
        loss = loss_sum if loss_sum is not None else ops.add(loss_a, loss_b)
        ops.stamp("gradients done")
        for grads in (g1, g2):
            for p, g in grads.items():
                if p.requires_grad:
                    p.grad = g.view(p.shape)
    return loss, l1, l2


def syn_train_step(model, x, labels, weights, scale_lamda, use_scale_term=True, mask=None, pooling="mean",
                   global_feature=None, before_cloak=None, combine=True, injected=None):
    """One forward + loss + backward of two_d_cnn_lstm_syn under the loss of training_cloak.py:133-149, scheduled by hand
    on the current stream: cloak -> frozen trunk -> cross-entropy (loss value + d logits in one kernel) -> the trunk's DATA
    gradient only (batch-summed where block 1 allows: the cloak's parameters are shared by the batch) -> one cloak backward
    kernel writing dL/dlocs, dL/drhos (with the scale term) into their flat slots.  `combine`: the 'combine*' datasets'
    loss sum_i w_i CE_i / B - scale_lamda log mean(scales) (:138-147); else the plain batch-mean cross-entropy (:149).
    `x` is (B, 1, H, W) or None with `before_cloak` producing it (the feature stage).  Returns (loss, logits)."""
    noise, net = model.intermed, model.original_model
    if net.pred == 'multitask':
        raise SeptError("two_d_cnn_lstm_syn training: pred='multitask' returns a tuple the reference's loss loop cannot "
                        "index either (training_cloak.py:141); use 'emotion' or 'gender'")
    head = 'emotion' if net.pred == 'emotion' else 'gender'
    injected = injected if injected is not None else getattr(model, "injected_masks", None)
    pool = "flatten" if pooling is None else "mean"
    locs, rhos = noise.locs, noise.rhos
    dev = rhos.device
    with torch.no_grad():
        ops.stamp("step start", dev)
        ops.begin_step(dev)
        eps = noise._epsilon(1)
        if before_cloak is not None:
            x = before_cloak()
        shape = x.shape
        B = shape[0]
        m = None if mask is None else mask.to(dev, torch.float32).contiguous()
        smin, smax = float(noise.min_scale), float(noise.max_scale)
        if isinstance(x, ops.LazyWindows):
            xn = ops.window_norm_cloak(x, locs.detach(), rhos.detach(), eps, m, smin, smax)
        else:
            xn = ops.cloak_forward(x.detach().float().contiguous().view(B, -1), locs.detach(), rhos.detach(), eps, m, smin, smax)
        xw = xn.view(B, shape[-2], shape[-1])
        need_dx = locs.requires_grad or rhos.requires_grad
        P = trunk_params(net, head, net.att)
        need_w = any(p.requires_grad for p in _param_list(P))
        logits, S = trunk_forward(xw, P, pool, need_grad=True, gfeat=global_feature, rng_site=SITE_EMOTION, injected=injected)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        d = ops.cross_entropy(logits, labels, weights if combine else None, 1.0 / B, loss)
        scale_mean = None
        if combine and use_scale_term and float(scale_lamda) != 0.0:
            _, scale_mean = ops.cloak_scales(rhos.detach(), smin, smax, want_scales=False, want_mean=True)
            ops.loss_sub_log(loss, scale_mean, float(scale_lamda))
        dx, grads = trunk_backward(S, P, d, need_wgrad=need_w, need_dx=need_dx, sum_dx=True) if (need_w or need_dx) else (None, {})
        if need_dx:
            da, _ = batch_sum_pair(dx, None)
            dlocs, drhos = ops.cloak_backward(da, None, 0.0, rhos.detach(), eps, m, smin, smax,
                                              scale_lambda=float(scale_lamda) if scale_mean is not None else 0.0,
                                              scale_mean=scale_mean, need_locs=locs.requires_grad, need_rhos=rhos.requires_grad,
                                              out_locs=grad_out(locs) if locs.requires_grad else None,
                                              out_rhos=grad_out(rhos) if rhos.requires_grad else None)
            if locs.requires_grad:
                locs.grad = dlocs
            if rhos.requires_grad:
                rhos.grad = drhos
        ops.stamp("gradients done")
        for p_, g in grads.items():
            if p_.requires_grad:
                p_.grad = g.view(p_.shape)
    return loss, logits


class ScalesFn(torch.autograd.Function):
    """cloak_noise.scales() (cloak_models.py:41-43)."""

    @staticmethod
    def forward(ctx, rhos, min_scale, max_scale):
        s, _ = ops.cloak_scales(rhos.detach(), min_scale, max_scale)
        ctx.save_for_backward(rhos.detach())
        ctx.cfg = (float(min_scale), float(max_scale))
        return s

    @staticmethod
    def backward(ctx, ds):
        (rhos,) = ctx.saved_tensors
        # d scales / d rho = (1 - tanh^2) / 2 * (max - min): reuse the cloak backward with eps = ds, B = 1
        ones = ds.contiguous().view(1, -1)
        _, dr = ops.cloak_backward(torch.ones_like(ones), None, 0.0, rhos, ds.contiguous(), None, ctx.cfg[0],
                                   ctx.cfg[1], need_locs=False)
        return dr, None, None


class GrlStepLossFn(torch.autograd.Function):
    """The loss of train() (training_cloak_with_grl.py:141-160) in two CE launches:
    sum_i w_i CE(emo_i)/B + gender_lambda * sum_i w_i CE(gen_i)/B - scale_lamda*log(mean(scales))."""

    @staticmethod
    def forward(ctx, preds, preds_grl, labels_emo, labels_gen, weights, gender_lambda, scale_lamda, rhos,
                min_scale, max_scale):
        B = preds.shape[0]
        loss = torch.empty((), dtype=torch.float32, device=preds.device)   # the first CE launch overwrites it
        d1 = ops.cross_entropy(preds.detach().contiguous(), labels_emo, weights, 1.0 / B, loss)
        d2 = None
        if preds_grl is not None:
            d2 = ops.cross_entropy(preds_grl.detach().contiguous(), labels_gen, weights, float(gender_lambda) / B,
                                   loss, accumulate=True)
        ctx.scale_cfg = None
        if rhos is not None and float(scale_lamda) != 0.0:
            _, mean = ops.cloak_scales(rhos.detach(), min_scale, max_scale, want_scales=False, want_mean=True)
            ops.loss_sub_log(loss, mean, float(scale_lamda))
            ctx.scale_cfg = (float(scale_lamda), float(min_scale), float(max_scale), mean)
            ctx.rhos = rhos.detach()
        ctx.d = (d1, d2)
        return loss

    @staticmethod
    def backward(ctx, g):
        d1, d2 = ctx.d
        # g is the upstream scalar (1.0 for loss.backward()); keep it on-device
        gd1 = ops.scale_dev(d1, g)
        gd2 = ops.scale_dev(d2, g) if d2 is not None else None
        drhos = None
        if ctx.scale_cfg is not None:
            lam, mn, mx, mean = ctx.scale_cfg
            rh = ctx.rhos
            zero = ops.fill(torch.empty((1, rh.numel()), dtype=torch.float32, device=rh.device), 0.0)
            _, drhos = ops.cloak_backward(zero, None, 0.0, rh, zero.view_as(rh), None, mn, mx, scale_lambda=lam,
                                          scale_mean=mean, need_locs=False)
            drhos = ops.scale_dev(drhos, g)
        return gd1, gd2, None, None, None, None, None, drhos, None, None


# ---------------------------------------------------------------------------------------------
# one_d_cnn_lstm (baseline_models.py:19-140): three Conv1d(k=5, pad=2)+ReLU+MaxPool1d+Dropout over
# time with the mel bins as channels, flatten, classifier Linear+ReLU+Dropout, head.  A pure CNN
# (its RNN is never called, :109).  Exact fp32: every product goes through sept_gemm.
# ---------------------------------------------------------------------------------------------
def one_d_forward(x, model, head, need_grad=True, injected=None):
    require_cuda(x)
    B, T, C = x.shape
    train = model.training
    inj = injected or {}
    convs = [m for m in model.conv if isinstance(m, torch.nn.Conv1d)]
    pools = [m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
             for m in model.conv if isinstance(m, torch.nn.MaxPool1d)]
    drops = [m.p for m in model.conv if isinstance(m, torch.nn.Dropout)]
    S = SimpleNamespace(layers=[], B=B, train=train)
    h = x.contiguous()
    for li, (cv, pool, dp) in enumerate(zip(convs, pools, drops)):
        cout, cin, _ = cv.weight.shape
        t = h.shape[1]
        wk = _cached("conv1d_w", cv.weight, lambda: ops.permute_cols(cv.weight.detach().reshape(cout, cin * 5), cin, 5))
        col = ops.unfold1d(h)
        y = ops.linear_forward(col, wk, cv.bias).view(B, t, cout)
        mask = None
        if train and (dp > 0 or "conv" in inj):
            m = inj.get("conv")
            mask = m[li] if m is not None else _drop_mask((B, t // pool, cout), x.device, dp)
        out, idx = ops.relu_pool1d_forward(y, pool, mask)
        S.layers.append(SimpleNamespace(col=col, y=y, idx=idx, mask=mask, pool=pool, wk=wk, cin=cin, cout=cout, t=t))
        h = out
    z = h.view(B, -1)
    lin = model.classifier[0]
    pdrop = model.classifier[2].p
    d1 = ops.linear_forward(z, lin.weight, lin.bias)
    dmask = None
    if train and (pdrop > 0 or "dense" in inj):
        m = inj.get("dense")
        dmask = m if m is not None else _drop_mask(tuple(d1.shape), x.device, pdrop)
    d1a = ops.relu_dropout_forward(d1, dmask)
    heads = list(head) if isinstance(head, (list, tuple)) else [head]   # pred='multitask': both, side by side
    if len(heads) == 1:
        logits = ops.linear_forward(d1a, head.weight, head.bias)
    else:
        logits = torch.empty((B, sum(h.weight.shape[0] for h in heads)), dtype=torch.float32, device=x.device)
        c0 = 0
        for h_ in heads:
            n = h_.weight.shape[0]
            ops.linear_forward(d1a, h_.weight, h_.bias, out=logits[:, c0:c0 + n])
            c0 += n
    S.z, S.d1, S.dmask, S.d1a = z, d1, dmask, d1a
    return logits, (S if need_grad else None)


def one_d_backward(S, model, head, dlogits, need_dx=False):
    grads = {}
    B = S.B
    convs = [m for m in model.conv if isinstance(m, torch.nn.Conv1d)]
    lin = model.classifier[0]
    dlogits = dlogits.contiguous()
    heads = list(head) if isinstance(head, (list, tuple)) else [head]
    d, c0 = None, 0
    for h_ in heads:
        n = h_.weight.shape[0]
        dl = dlogits[:, c0:c0 + n]
        c0 += n
        if d is None:
            d = ops.linear_backward_input(dl, h_.weight)
        else:   # second head of pred='multitask': accumulate
            ops.gemm_raw(dl, dl.stride(0), 1, h_.weight, h_.weight.shape[1], 1, d, d.stride(0), B,
                         h_.weight.shape[1], n, beta=1.0)
        grads[h_.weight] = ops.linear_backward_weight(dl, S.d1a)
        grads[h_.bias] = ops.colsum(dl)
    d = ops.relu_dropout_backward(d, S.d1, S.dmask)
    grads[lin.weight] = ops.linear_backward_weight(d, S.z)
    grads[lin.bias] = ops.colsum(d)
    dh = ops.linear_backward_input(d, lin.weight)
    for li in range(len(S.layers) - 1, -1, -1):
        L, cv = S.layers[li], convs[li]
        dh = dh.view(B, L.t // L.pool, L.cout)
        dy = ops.relu_pool1d_backward(dh.contiguous(), L.y, L.idx, L.pool, L.mask).view(B * L.t, L.cout)
        dwk = ops.linear_backward_weight(dy, L.col)
        grads[cv.weight] = ops.permute_cols(dwk, L.cin, 5, inverse=True).view(L.cout, L.cin, 5)
        grads[cv.bias] = ops.colsum(dy)
        if li > 0 or need_dx:
            dcol = ops.linear_backward_input(dy, L.wk)
            dh = ops.fold1d(dcol, B, L.t, L.cin)
    return (dh if need_dx else None), grads


class OneDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, head, injected, *params):
        need = any(ctx.needs_input_grad)
        logits, S = one_d_forward(x.detach().float().contiguous(), model, head, need_grad=need, injected=injected)
        ctx.S, ctx.model, ctx.head, ctx.params = S, model, head, params
        ctx.need_dx = x.requires_grad
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        dx, grads = one_d_backward(ctx.S, ctx.model, ctx.head, dlogits, need_dx=ctx.need_dx)
        ctx.S = None
        return (dx, None, None, None) + tuple(grads.get(p) if p.requires_grad else None for p in ctx.params)


def run_one_d(model, x, head, injected=None):
    params = [p for m in model.conv if isinstance(m, torch.nn.Conv1d) for p in (m.weight, m.bias)]
    params += [model.classifier[0].weight, model.classifier[0].bias]
    for h_ in (head if isinstance(head, (list, tuple)) else [head]):
        params += [h_.weight, h_.bias]
    return OneDFn.apply(x, model, head, injected, *params)
