"""What a 16-bit ConvLSTM_w_ref pipeline can be expected to deliver: the reference network in float64 arithmetic with
round-to-nearest-even to bf16 (or fp16) at exactly the places where k_fused.hip / k_lstm_x16.hip hold 16-bit values.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py header for who may import this).  Used by the GPU parity tests to tell the
kernel's own error from the arithmetic's: a correct 16-bit kernel lands on the emulation's error level, not below it.

Rounding sites (models/ConvLSTM_w_ref.py:41-56 with BatchNorm folded as model_util.py:199-221 does):
  wconv  : weights of the five matrix-core convolutions (sig_conv2/3, seq_conv1/2, merge_conv1); sig_conv1 is fp32 VALU
  aconv  : activations between the convolutions (sig1, sig2, seq1, cat)
  x      : merge_conv1's output, the LSTM input
  wlstm  : W_ih / W_hh of lstm1, W_ih of lstm2
  h      : h_t before it feeds step t+1 / lstm2 (the cell state and all accumulations stay fp32 in the kernels, float64 here)
A site may also be named with its layer ("wconv.merge1", "aconv.cat"); `split` lists sites that keep two 16-bit parts
(hi + lo) instead of one."""
import torch

ALL_SITES = ("wconv", "aconv", "x", "wlstm", "h")
_FMT = {"bf16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16}


def _round(t, fmt):
    return t.to(torch.float32).to(fmt).to(torch.float64)


def _fold(net, conv, bn):
    c, b = getattr(net, conv), getattr(net, bn)
    s = b.weight.double() / torch.sqrt(b.running_var.double() + b.eps)
    return c.weight.double() * s[:, None, None], (c.bias.double() - b.running_mean.double()) * s + b.bias.double()


def _swish(x):
    return x * torch.sigmoid(x)


def forward(net, sig, enc, sites=ALL_SITES, split=(), fmt="bf16", prescale=1.0):
    """net: oracle.torch_ref.ConvLSTMRef; sig [n,1,L], enc [n,4K,L] (any float dtype) -> float64 logits [n,num_out].
    sites=() is the exact float64 evaluation.  `prescale`: every value is rounded as round(v * prescale) / prescale - the same
    relative rounding unit, another DRAW of the rounding errors (the kernels round weights that carry a gate pre-scale, for
    example).  On the amplified synthetic networks the mean error of one draw spreads by 2-3 x between draws (sizes 96 / 128,
    1500 chunks: 1.3e-3 .. 4.5e-3), so a gate on a kernel's error compares with the envelope of a few draws, not with one."""
    fmt = _FMT[fmt]
    sig, enc = sig.double(), enc.double()
    _rnd1 = (lambda t, f: _round(t * prescale, f) / prescale) if prescale != 1.0 else _round

    def rnd(name, t, sub=None):
        on = name in sites or (sub is not None and f"{name}.{sub}" in sites)
        if not on:
            return t
        if name in split or (sub is not None and f"{name}.{sub}" in split):
            hi = _rnd1(t, fmt)
            return hi + _rnd1(t - hi, fmt)
        return _rnd1(t, fmt)

    F = torch.nn.functional
    conv = lambda x, wb, sub, stride=1: F.conv1d(x, rnd("wconv", wb[0], sub), wb[1], stride=stride)  # noqa: E731
    s = _swish(F.conv1d(sig, *_fold(net, "sig_conv1", "sig_bn1")))
    s = _swish(conv(rnd("aconv", s, "sig1"), _fold(net, "sig_conv2", "sig_bn2"), "sig2"))
    s = _swish(conv(rnd("aconv", s, "sig2"), _fold(net, "sig_conv3", "sig_bn3"), "sig3", 3))
    q = _swish(conv(enc, _fold(net, "seq_conv1", "seq_bn1"), "seq1"))
    q = _swish(conv(rnd("aconv", q, "seq1"), _fold(net, "seq_conv2", "seq_bn2"), "seq2", 3))
    z = rnd("aconv", torch.cat((s, q), 1), "cat")
    x = rnd("x", _swish(conv(z, _fold(net, "merge_conv1", "merge_bn"), "merge1"))).permute(2, 0, 1)  # [T][n][size]
    wih, whh = rnd("wlstm", net.lstm1.weight_ih_l0.double()), rnd("wlstm", net.lstm1.weight_hh_l0.double())
    b = net.lstm1.bias_ih_l0.double() + net.lstm1.bias_hh_l0.double()
    n, size = x.shape[1], x.shape[2]
    h = torch.zeros(n, size, dtype=torch.float64)
    c = torch.zeros(n, size, dtype=torch.float64)
    for t in range(x.shape[0]):
        g = x[t] @ wih.T + rnd("h", h) @ whh.T + b
        i, f, gg, o = g.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
    # after the two flips z[-1] is ONE lstm2 step on swish(h1[T-1]) with zero state (ConvLSTM_w_ref.py:53-54)
    g = rnd("h", _swish(h)) @ rnd("wlstm", net.lstm2.weight_ih_l0.double()).T + net.lstm2.bias_ih_l0.double() + net.lstm2.bias_hh_l0.double()
    i, f, gg, o = g.chunk(4, 1)
    y = _swish(torch.sigmoid(o) * torch.tanh(torch.sigmoid(i) * torch.tanh(gg)))
    return y @ net.fc.weight.double().T + net.fc.bias.double()
