"""CPU oracle for the model half of the hot path (TEST INFRASTRUCTURE ONLY).

A plain-torch (fp32, autograd) restatement of the reference modules that sit on the path:
    GradientReversalFunction / GradientReversal      model/reversal_gradient.py:5-32
    cloak_noise                                      model/cloak_models.py:24-58
    two_d_cnn_lstm (+ deep_two_d_cnn_lstm, its LSTM clone deep_two_d_cnn_lstm_tmp, one_d_cnn_lstm)   model/baseline_models.py:19-509
    two_d_cnn_lstm_syn / two_d_cnn_lstm_syn_with_grl model/cloak_models.py:61-226
    the per-step loss of train()                     training/training_cloak_with_grl.py:138-160
with the reference's constructor signatures, attribute names and state-dict keys, so a
state dict moves between the reference modules, this oracle and the HIP-backed product
modules unchanged.

Pinned by tests/golden/model_golden.npz, produced by tools/make_goldens_model.py which
imports the REFERENCE modules from /root/reference/model in the build container, loads
closed-form weights into them and records their outputs/gradients; tests/test_oracle_model.py
checks this restatement against those vectors.

Differences from the reference that are deliberate and test-only:
  * the cloak epsilon can be injected (``cloak_noise.eps``) and so can every dropout mask of a
    two_d_cnn_lstm-style network (``model.drop = {'drop2d': [(B, C) per conv block], 'rnn': (B, T, 2H),
    'dense': (B, 128)}`` -- SCALE masks: 0 or 1 / (1 - p), i.e. what nn.Dropout multiplies by), so that train mode
    is reproducible across implementations.  Dropout2d drops whole channels of a sample (mask (B, C));
    nn.GRU / nn.LSTM apply their dropout to the output of every layer but the last (ATen RNN.cpp,
    apply_layer_stack), so with an 'rnn' mask the two-layer module is run as two one-layer calls on the same
    weights with the mask in between (`_rnn_layers`).  With nothing injected the modules behave like the
    reference (torch RNG).  tests/golden/model_golden_step.npz pins this hook: tools/make_goldens_step.py
    reads the masks the REFERENCE drew in a train-mode step (forward hooks on its Dropout modules; the
    recurrent mask by replaying the generator state in front of its nn.GRU call) and records the reference's
    outputs and gradients under them.
  * ``sim_bf16`` (set by ``simulate_bf16(model)``): the conv stack rounds to bf16 exactly where the HIP path
    stores bf16 -- conv weights of the 5x5 MFMA layers, the pre-BatchNorm conv outputs and the pooled block
    outputs in the forward pass, and the gradients of those two activation tensors in the backward pass -- so
    that both sides take the SAME max-pool decisions and a gradient comparison is not blurred by near-ties
    (an fp32 network and a bf16 one pick different, equally valid window maxima).  Off by default: the
    restatement pinned to the reference is the fp32 one.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# gradient reversal (reversal_gradient.py:5-32)
# ----------------------------------------------------------------------------------------
class GradientReversalFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lambda_):
        ctx.lambda_ = lambda_
        return x.clone()

    @staticmethod
    def backward(ctx, grads):
        return -ctx.lambda_ * grads, None


class GradientReversal(nn.Module):
    def __init__(self, lambda_=1):
        super().__init__()
        self.lambda_ = lambda_

    def forward(self, x):
        return GradientReversalFunction.apply(x, self.lambda_)


# ----------------------------------------------------------------------------------------
# cloak noise (cloak_models.py:24-58)
# ----------------------------------------------------------------------------------------
class cloak_noise(nn.Module):
    def __init__(self, given_locs, given_scales, min_scale, max_scale, device):
        super().__init__()
        size = given_scales.shape
        self.min_scale, self.max_scale = min_scale, max_scale
        self.given_locs, self.given_scales = given_locs, given_scales
        self.locs = nn.Parameter(torch.Tensor(size).copy_(given_locs))
        self.rhos = nn.Parameter(torch.ones(size) - 3)
        self.device = device
        self.normal = torch.distributions.normal.Normal(0, 0.1)
        self.eps = None  # test hook: injected epsilon

    def scales(self):
        return (1.0 + torch.tanh(self.rhos)) / 2 * (self.max_scale - self.min_scale) + self.min_scale

    def sample_noise(self, mask=None):
        eps = self.eps if self.eps is not None else self.normal.sample(self.rhos.shape).to(self.device)
        if mask is not None:
            eps = eps * mask
        return self.locs + self.scales() * eps

    def forward(self, input, mask=None):
        noise = self.sample_noise(mask)
        return input + noise if mask is None else input * mask + noise


# ----------------------------------------------------------------------------------------
# bf16 storage simulation (test hook, see the header)
# ----------------------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
    """A tensor the HIP path keeps in bf16: value rounded forward, its gradient rounded backward."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _round_weight(w):
    """bf16 operand copy of an fp32 master weight (the gradient reaches the master unrounded)."""
    return w + (w.detach().bfloat16().float() - w.detach())


def simulate_bf16(model, on=True):
    """Switch the bf16 storage simulation of every two_d_cnn_lstm-style network inside `model`."""
    for m in model.modules():
        if isinstance(m, _TwoDBase):
            m.sim_bf16 = bool(on)
    return model


# ----------------------------------------------------------------------------------------
# baseline classifiers (baseline_models.py)
# ----------------------------------------------------------------------------------------
def _rnn_cell(name):
    if name.lower() == "lstm":
        return nn.LSTM
    if name.lower() == "gru":
        return nn.GRU
    raise ValueError("Unsupported RNN Cell: {0}".format(name))


def _conv_block(cin, cout, p, pool=True):
    layers = [nn.Conv2d(cin, cout, kernel_size=5, padding=2), nn.BatchNorm2d(cout), nn.ReLU()]
    if pool:
        layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
    layers.append(nn.Dropout2d(p))
    return layers


class _TwoDBase(nn.Module):
    """Shared body of two_d_cnn_lstm (baseline_models.py:143-260) and deep_two_d_cnn_lstm
    (:264-385).  `deep` adds a 4th conv without pooling and flattens instead of averaging."""

    def __init__(self, deep, input_channel, input_spec_size, cnn_filter_size, lstm_hidden_size=128,
                 num_layers_lstm=2, pred="emotion", bidirectional=True, rnn_cell="gru", attention_size=256,
                 variable_lengths=False, global_feature=1, att=None):
        super().__init__()
        self.input_channel, self.input_spec_size = input_channel, input_spec_size
        self.lstm_hidden_size, self.bidirectional = lstm_hidden_size, bidirectional
        self.num_layers_lstm, self.dropout_p = num_layers_lstm, 0.2
        self.variable_lengths = variable_lengths
        self.num_emo_classes, self.num_gender_class = 4, 2
        self.cnn_filter_size, self.attention_size = cnn_filter_size, attention_size
        self.pred, self.att = pred, att
        self.deep = deep
        self.sim_bf16 = False
        self.drop = None   # test hook: explicit dropout scale masks (see the header)
        self.rnn_input_size = int(128 * input_spec_size / 8)
        self.rnn_cell = _rnn_cell(rnn_cell)
        self.dropout = nn.Dropout(p=self.dropout_p)
        layers = _conv_block(1, 32, self.dropout_p) + _conv_block(32, 64, self.dropout_p) + \
            _conv_block(64, 128, self.dropout_p)
        if deep:
            layers += _conv_block(128, 128, self.dropout_p, pool=False)
        self.conv = nn.Sequential(*layers)
        self.rnn = self.rnn_cell(input_size=self.rnn_input_size, hidden_size=lstm_hidden_size,
                                 num_layers=num_layers_lstm, batch_first=True, dropout=self.dropout_p,
                                 bidirectional=bidirectional)
        d_att, n_att = attention_size, 16
        self.att_linear1 = nn.Linear(lstm_hidden_size * 2, d_att, bias=False)
        self.att_pool = nn.Tanh()
        self.att_linear2 = nn.Linear(d_att, n_att, bias=False)
        self.att_mat1 = nn.Parameter(torch.rand(d_att, lstm_hidden_size * 2))
        self.att_mat2 = nn.Parameter(torch.rand(n_att, d_att))
        self.dense_relu1, self.dense_relu2 = nn.ReLU(), nn.ReLU()
        self.dense2 = nn.Linear(128, 64)
        if global_feature == 1:
            self.dense1 = nn.Linear(lstm_hidden_size * 2 + 88, 128)
        else:
            self.dense1 = nn.Linear(lstm_hidden_size * 2 * (25 if deep else 1), 128)
        self.pred_emotion_layer = nn.Linear(128, self.num_emo_classes)
        self.pred_gender_layer = nn.Linear(128, self.num_gender_class)
        # reference init_weight() iterates the *names* in self._modules, so it changes
        # nothing: weights stay at torch default init.

    # the trunk shared with the cloak wrappers (cloak_models.py:165-193)
    def _injected(self, key):
        return self.drop.get(key) if (self.training and self.drop is not None) else None

    def _conv_blocks(self, x):
        """self.conv block by block: with `sim_bf16` the HIP path's bf16 storage points (conv1 runs on split-bf16
        operands ~ fp32); with an injected 'drop2d' list the Dropout2d modules are replaced by the given (B, C) masks."""
        mods = list(self.conv)
        if mods and isinstance(mods[0], GradientReversal):      # Sequential(GradientReversal, conv) of the GRL wrapper
            x, mods = mods[0](x), list(mods[1])
        sim, d2 = self.sim_bf16, self._injected("drop2d")
        i, blk = 0, 0
        while i < len(mods):
            conv, bn, j = mods[i], mods[i + 1], i + 3           # Conv2d, BatchNorm2d, ReLU
            w = _round_weight(conv.weight) if (sim and blk > 0) else conv.weight
            pre = F.conv2d(x, w, conv.bias, padding=2)
            y = F.relu(bn(_RoundBF16.apply(pre) if sim else pre))
            if isinstance(mods[j], nn.MaxPool2d):
                y, j = mods[j](y), j + 1
            y = mods[j](y) if d2 is None else y * d2[blk].to(y.dtype)[:, :, None, None]     # Dropout2d
            x = _RoundBF16.apply(y) if sim else y               # the stored block output
            i, blk = j + 1, blk + 1
        return x

    def _rnn_layers(self, x):
        """self.rnn(x)[0]; with an injected 'rnn' mask: layer 0, the mask, layer 1 (what ATen does with its own draw)."""
        m = self._injected("rnn")
        if m is None:
            return self.rnn(x)[0]
        r = self.rnn
        assert r.num_layers == 2 and r.bidirectional and r.batch_first
        lstm = isinstance(r, nn.LSTM)
        for layer in range(2):
            flat = [getattr(r, f"{n}_l{layer}{sfx}") for sfx in ("", "_reverse")
                    for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            h0 = x.new_zeros(2, x.shape[0], r.hidden_size)
            if lstm:
                x = torch._VF.lstm(x, (h0, h0), flat, True, 1, 0.0, self.training, True, True)[0]
            else:
                x = torch._VF.gru(x, h0, flat, True, 1, 0.0, self.training, True, True)[0]
            if layer == 0:
                x = x * m.to(x.dtype)
        return x

    def features(self, x, global_feature=None, pooling="model"):
        x = self._conv_blocks(x.float()) if (self.sim_bf16 or self._injected("drop2d") is not None) else self.conv(x.float())
        x = x.transpose(1, 2).contiguous()
        s = x.size()
        x = x.reshape(-1, s[1], s[2] * s[3])
        x = self._rnn_layers(x)
        if self.att is None:
            flatten = self.deep if pooling == "model" else (pooling is None)
            z = x.reshape(-1, x.size(1) * x.size(2)) if flatten else torch.mean(x, dim=1)
        elif self.att == "self_att":
            a = self.att_linear2(self.att_pool(self.att_linear1(x))).transpose(1, 2)
            z = torch.mean(torch.matmul(torch.softmax(a, dim=2), x), dim=1)
        if global_feature is not None:
            z = torch.cat((z, global_feature), 1)
        z = self.dense_relu1(self.dense1(z))
        dm = self._injected("dense")
        return self.dropout(z) if dm is None else z * dm.to(z.dtype)

    def head(self, z):
        if self.pred == "multitask":
            return self.pred_emotion_layer(z), self.pred_gender_layer(z)
        if self.pred == "emotion":
            return self.pred_emotion_layer(z)
        return self.pred_gender_layer(z)

    def forward(self, input_var, global_feature=None):
        return self.head(self.features(input_var.float(), global_feature))


class two_d_cnn_lstm(_TwoDBase):
    def __init__(self, *a, **k):
        super().__init__(False, *a, **k)


class deep_two_d_cnn_lstm(_TwoDBase):
    def __init__(self, *a, **k):
        super().__init__(True, *a, **k)


class deep_two_d_cnn_lstm_tmp(_TwoDBase):
    """baseline_models.py:388-509: the deep variant again, with rnn_cell defaulting to 'lstm'."""

    def __init__(self, input_channel, input_spec_size, cnn_filter_size, lstm_hidden_size=128, num_layers_lstm=2,
                 pred="emotion", bidirectional=True, rnn_cell="lstm", attention_size=256, variable_lengths=False,
                 global_feature=1, att=None):
        super().__init__(True, input_channel, input_spec_size, cnn_filter_size, lstm_hidden_size, num_layers_lstm, pred,
                         bidirectional, rnn_cell, attention_size, variable_lengths, global_feature, att)


class one_d_cnn_lstm(nn.Module):
    """baseline_models.py:19-140 -- a pure CNN: the RNN is constructed but never called."""

    def __init__(self, input_channel, input_spec_size, cnn_filter_size, lstm_hidden_size=128, num_layers_lstm=2,
                 pred="emotion", bidirectional=True, rnn_cell="gru", attention_size=256,
                 variable_lengths=False, global_feature=1, att=None):
        super().__init__()
        self.input_spec_size, self.lstm_hidden_size = input_spec_size, lstm_hidden_size
        self.dropout_p, self.pred, self.att = 0.2, pred, att
        self.rnn_input_size = 512
        p = self.dropout_p
        self.dropout = nn.Dropout(p=p)
        self.conv = nn.Sequential(
            nn.Conv1d(input_spec_size, 128, kernel_size=5, padding=2), nn.ReLU(), nn.MaxPool1d(2, 2), nn.Dropout(p),
            nn.Conv1d(128, 256, kernel_size=5, padding=2), nn.ReLU(), nn.MaxPool1d(5, 5), nn.Dropout(p),
            nn.Conv1d(256, 512, kernel_size=5, padding=2), nn.ReLU(), nn.MaxPool1d(5, 5), nn.Dropout(p))
        self.rnn = _rnn_cell(rnn_cell)(input_size=512, hidden_size=lstm_hidden_size, num_layers=num_layers_lstm,
                                       batch_first=True, dropout=p, bidirectional=bidirectional)
        d_att, n_att = attention_size, 8
        self.att_linear1 = nn.Linear(lstm_hidden_size * 2, d_att)
        self.att_pool = nn.Tanh()
        self.att_linear2 = nn.Linear(d_att, n_att)
        self.att_mat1 = nn.Parameter(torch.rand(d_att, lstm_hidden_size * 2))
        self.att_mat2 = nn.Parameter(torch.rand(n_att, d_att))
        self.dense_relu1, self.dense_relu2 = nn.ReLU(), nn.ReLU()
        self.classifier = nn.Sequential(nn.Linear(512 * 4, 128), nn.ReLU(), nn.Dropout(p))
        self.dense2 = nn.Linear(128, 64)
        self.dense1 = nn.Linear(lstm_hidden_size * 2 + 88, 128) if global_feature == 1 else nn.Linear(512 * 4, 128)
        self.pred_emotion_layer = nn.Linear(128, 4)
        self.pred_gender_layer = nn.Linear(128, 2)

    def forward(self, input_var, global_feature=None):
        x = input_var.squeeze(dim=1).permute(0, 2, 1)
        x = self.conv(x.float()).permute(0, 2, 1)
        z = x.reshape(-1, x.size(1) * x.size(2))
        if global_feature is not None:
            z = torch.cat((z, global_feature), 1)
        z = self.classifier(z)
        if self.pred == "multitask":
            return self.pred_emotion_layer(z), self.pred_gender_layer(z)
        return self.pred_emotion_layer(z) if self.pred == "emotion" else self.pred_gender_layer(z)


# ----------------------------------------------------------------------------------------
# cloak wrappers (cloak_models.py:61-226)
# ----------------------------------------------------------------------------------------
def _freeze(model):
    # cloak_models.py:69-76 / 142-149: only requires_grad is cleared; BatchNorm and Dropout
    # stay in whatever mode .train()/.eval() puts them.
    for p in model.parameters():
        p.requires_grad = False


class two_d_cnn_lstm_syn(nn.Module):
    def __init__(self, original_model, noise_model):
        super().__init__()
        self.intermed, self.original_model = noise_model, original_model
        _freeze(original_model)

    def forward(self, input_var, global_feature=None, mask=None, pooling=None):
        x = input_var.float()
        x = self.intermed(x) if mask is None else self.intermed(x, mask)
        noisy = x.detach()
        m = self.original_model
        z = m.features(x, global_feature, pooling)
        return m.head(z), noisy


class two_d_cnn_lstm_syn_with_grl(nn.Module):
    def __init__(self, original_model, gender_model, noise_model, grl_lambda):
        super().__init__()
        self.intermed, self.original_model, self.gender_model = noise_model, original_model, gender_model
        _freeze(original_model)
        self.gender_model.conv = nn.Sequential(GradientReversal(grl_lambda), gender_model.conv)

    def forward(self, input_var, global_feature=None, mask=None, grl=False, pooling=None):
        x = input_var.float()
        x = self.intermed(x) if mask is None else self.intermed(x, mask)
        noisy = x.detach()
        z1 = self.original_model.features(x, global_feature, pooling)
        preds1 = self.original_model.pred_emotion_layer(z1)
        z2 = self.gender_model.features(x, global_feature, pooling)
        preds2 = self.gender_model.pred_gender_layer(z2)
        return preds1, preds2, noisy


# ----------------------------------------------------------------------------------------
# the step loss (training_cloak_with_grl.py:141-160)
# ----------------------------------------------------------------------------------------
def grl_step_loss(preds, preds_grl, labels_emo, labels_gen, weights, gender_lambda, scale_lamda,
                  cloak_model, training=True, suppression=False):
    """sum_i w_i CE(emo_i)/B + gender_lambda * sum_i w_i CE(gen_i)/B - scale_lamda*log(mean(scales)).
    `weights` (B,) per-sample speaker weights (ignored in validate mode, :153-154)."""
    B = preds.shape[0]
    ce_e = F.cross_entropy(preds, labels_emo.view(-1), reduction="none")
    ce_g = F.cross_entropy(preds_grl, labels_gen.view(-1), reduction="none")
    w = weights if training else torch.ones_like(ce_e)
    total = (ce_e * w).sum() / B + float(gender_lambda) * (ce_g * w).sum() / B
    if not suppression:
        total = total - float(scale_lamda) * torch.log(torch.mean(cloak_model.intermed.scales()))
    return total


def syn_step_loss(preds, labels, weights, scale_lamda, cloak_model, training=True, suppression=False, combine=True):
    """The loss of training_cloak.py:137-149 for two_d_cnn_lstm_syn.  `combine` (the args.dataset 'combine*' branch,
    :138-147): sum_i w_i CE_i / B (w dropped in validate mode) - scale_lamda * log(mean(scales)) unless suppressing;
    otherwise (:149) the plain mean cross-entropy of the batch."""
    if not combine:
        return F.cross_entropy(preds, labels.view(-1))
    ce = F.cross_entropy(preds, labels.view(-1), reduction="none")
    w = weights if (training and weights is not None) else torch.ones_like(ce)
    total = (ce * w).sum() / preds.shape[0]
    if not suppression:
        total = total - float(scale_lamda) * torch.log(torch.mean(cloak_model.intermed.scales()))
    return total
