"""torch.nn restatement of the reference networks, for the CPU-PyTorch baseline timing and
as a second, independent checker of logits.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py header for who may import this).

Restates models/ConvLSTM_w_ref.py:11-58 and models/Conv_w_ref.py:11-62 with the identical
ATen ops (Conv1d, BatchNorm1d eval, LSTM, Linear, x*sigmoid(x)) and identical parameter
names, so a reference state_dict loads with strict=True.  Pinned by
tests/test_oracle_golden.py against reference logits.
"""
import torch
from torch import nn


def _swish(x):
    # src/remora/activations.py:4-18
    return x * torch.sigmoid(x)


class ConvLSTMRef(nn.Module):
    def __init__(self, size=64, kmer_len=9, num_out=2):
        super().__init__()
        self.sig_conv1, self.sig_bn1 = nn.Conv1d(1, 4, 5), nn.BatchNorm1d(4)
        self.sig_conv2, self.sig_bn2 = nn.Conv1d(4, 16, 5), nn.BatchNorm1d(16)
        self.sig_conv3, self.sig_bn3 = nn.Conv1d(16, size, 9, 3), nn.BatchNorm1d(size)
        self.seq_conv1, self.seq_bn1 = nn.Conv1d(kmer_len * 4, 16, 5), nn.BatchNorm1d(16)
        self.seq_conv2, self.seq_bn2 = nn.Conv1d(16, size, 13, 3), nn.BatchNorm1d(size)
        self.merge_conv1, self.merge_bn = nn.Conv1d(size * 2, size, 5), nn.BatchNorm1d(size)
        self.lstm1 = nn.LSTM(size, size, 1)
        self.lstm2 = nn.LSTM(size, size, 1)
        self.fc = nn.Linear(size, num_out)

    def forward(self, sigs, seqs):
        s = _swish(self.sig_bn1(self.sig_conv1(sigs)))
        s = _swish(self.sig_bn2(self.sig_conv2(s)))
        s = _swish(self.sig_bn3(self.sig_conv3(s)))
        q = _swish(self.seq_bn1(self.seq_conv1(seqs)))
        q = _swish(self.seq_bn2(self.seq_conv2(q)))
        z = torch.cat((s, q), 1)
        z = _swish(self.merge_bn(self.merge_conv1(z)))
        z = z.permute(2, 0, 1)
        z = _swish(self.lstm1(z)[0])
        z = torch.flip(_swish(self.lstm2(torch.flip(z, (0,)))[0]), (0,))
        return self.fc(z[-1])


class ConvOnlyRef(nn.Module):
    def __init__(self, size=64, kmer_len=9, num_out=2):
        super().__init__()
        self.sig_conv1, self.sig_bn1 = nn.Conv1d(1, 4, 11), nn.BatchNorm1d(4)
        self.sig_conv2, self.sig_bn2 = nn.Conv1d(4, 16, 11), nn.BatchNorm1d(16)
        self.sig_conv3, self.sig_bn3 = nn.Conv1d(16, size, 9, 3), nn.BatchNorm1d(size)
        self.seq_conv1, self.seq_bn1 = nn.Conv1d(kmer_len * 4, 16, 11), nn.BatchNorm1d(16)
        self.seq_conv2, self.seq_bn2 = nn.Conv1d(16, 32, 11), nn.BatchNorm1d(32)
        self.seq_conv3, self.seq_bn3 = nn.Conv1d(32, size, 9, 3), nn.BatchNorm1d(size)
        self.merge_conv1, self.merge_bn1 = nn.Conv1d(size * 2, size, 5), nn.BatchNorm1d(size)
        self.merge_conv2, self.merge_bn2 = nn.Conv1d(size, size, 5), nn.BatchNorm1d(size)
        self.merge_conv3, self.merge_bn3 = nn.Conv1d(size, size, 3, 2), nn.BatchNorm1d(size)
        self.merge_conv4, self.merge_bn4 = nn.Conv1d(size, size, 3, 2), nn.BatchNorm1d(size)
        self.fc = nn.Linear(size * 3, num_out)

    def forward(self, sigs, seqs):
        s = _swish(self.sig_bn1(self.sig_conv1(sigs)))
        s = _swish(self.sig_bn2(self.sig_conv2(s)))
        s = _swish(self.sig_bn3(self.sig_conv3(s)))
        q = _swish(self.seq_bn1(self.seq_conv1(seqs)))
        q = _swish(self.seq_bn2(self.seq_conv2(q)))
        q = _swish(self.seq_bn3(self.seq_conv3(q)))
        z = torch.cat((s, q), 1)
        z = _swish(self.merge_bn1(self.merge_conv1(z)))
        z = _swish(self.merge_bn2(self.merge_conv2(z)))
        z = _swish(self.merge_bn3(self.merge_conv3(z)))
        z = _swish(self.merge_bn4(self.merge_conv4(z)))
        return self.fc(torch.flatten(z, start_dim=1))


def build(arch, size=64, kmer_len=9, num_out=2):
    cls = {"conv_lstm": ConvLSTMRef, "conv_only": ConvOnlyRef}[arch]
    return cls(size=size, kmer_len=kmer_len, num_out=num_out).eval()


def random_model(arch="conv_lstm", size=64, kmer_len=9, num_out=2, seed=0):
    """SURVEY §8(d) weights: torch.manual_seed init + randomised BN running stats."""
    torch.manual_seed(seed)
    net = build(arch, size, kmer_len, num_out)
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, nn.BatchNorm1d):
                n = mod.num_features
                mod.running_mean.copy_(torch.randn(n, generator=gen))
                mod.running_var.copy_(torch.rand(n, generator=gen) * 1.5 + 0.5)
                mod.weight.copy_(1.0 + 0.2 * torch.randn(n, generator=gen))
                mod.bias.copy_(0.2 * torch.randn(n, generator=gen))
        for pname, p in net.named_parameters():
            if pname.startswith("lstm") and "weight" in pname:
                p.mul_(2.5)
            if pname.startswith("fc."):
                p.mul_(12.0 if hasattr(net, "lstm1") else 1.5)  # keep logits within a few units
            if pname.startswith("lstm") and pname.endswith("bias_ih_l0"):
                n4 = p.shape[0] // 4
                p[n4 : 2 * n4] += 2.0  # forget-gate bias: longer memory, as in trained LSTMs
            if "conv" in pname and pname.endswith("weight"):
                p.mul_(2.45)  # He-scale: keeps the logits sensitive to the inputs
    for p in net.parameters():
        p.requires_grad = False
    return net.eval()


def from_state(state):
    """state: {torch-name: np.ndarray} -> eval module (strict load)."""
    arch = "conv_lstm" if any(k.startswith("lstm1") for k in state) else "conv_only"
    size = state["sig_conv3.weight"].shape[0]
    kmer_len = state["seq_conv1.weight"].shape[1] // 4
    num_out = state["fc.weight"].shape[0]
    net = build(arch, size, kmer_len, num_out)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()}, strict=True)
    for p in net.parameters():
        p.requires_grad = False
    return net.eval()
