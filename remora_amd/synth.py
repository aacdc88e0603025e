"""Synthetic workloads of SURVEY.md §8(d): chunk arrays in the CoreRemoraDataset layout and
synthetic reads.  numpy only; seeded; used by bench.py and the tests."""
import numpy as np

CONFIGS = {
    # name: (chunk_context, kmer_context_bases, max_seq_len, num_out, cg_context)
    "C100": ((50, 50), (4, 4), 20, 2, True),
    "C200": ((100, 100), (4, 4), 40, 3, False),
}


def synth_chunks(n, chunk_len=100, max_seq_len=20, kmer_context_bases=(4, 4), num_out=2, cg_context=True,
                 seed=20240926, shard=0, block=1 << 16):
    """n chunks: seq_len ~ U[ceil(0.3*max), max]; mapping = sorted distinct cuts in (0, L);
    bases uniform with the C(G) context forced at the base covering L/2; signal ~ N(0,1);
    labels uniform.  Vectorised restatement of the per-chunk recipe (same distributions)."""
    rng = np.random.default_rng(seed + shard)
    kb, ka = kmer_context_bases
    L = chunk_len
    lo = max(2, int(np.ceil(0.3 * max_seq_len)))
    seq_w, map_w = max_seq_len + kb + ka, max_seq_len + 1
    signal = np.empty((n, 1, L), np.float32)
    seqs = np.full((n, seq_w), -1, np.int8)
    maps = np.zeros((n, map_w), np.int16)
    lens = np.empty(n, np.int16)
    for st in range(0, n, block):
        m = min(block, n - st)
        sl = rng.integers(lo, max_seq_len + 1, m)
        # sl-1 distinct cut points out of 1..L-1: rank random keys, keep the sl-1 smallest
        keys = rng.random((m, L - 1), dtype=np.float32)
        order = np.argsort(keys, axis=1)[:, : max_seq_len - 1] + 1   # candidate cuts
        col = np.arange(max_seq_len - 1)[None, :]
        cuts = np.where(col < (sl - 1)[:, None], order, L + 1)        # unused -> sentinel
        cuts.sort(axis=1)
        mp = np.zeros((m, map_w), np.int64)
        mp[:, 1:max_seq_len] = cuts
        mp[mp > L] = 0
        mp[np.arange(m), sl] = L
        maps[st : st + m] = mp
        lens[st : st + m] = sl
        bases = rng.integers(0, 4, (m, seq_w)).astype(np.int8)
        valid = np.arange(seq_w)[None, :] < (sl + kb + ka)[:, None]
        bases[~valid] = -1
        # focus base p* = the base whose [map[p], map[p+1]) contains L//2
        inside = (mp[:, :-1] <= L // 2) & (np.arange(map_w - 1)[None, :] < sl[:, None])
        pstar = inside.sum(axis=1) - 1
        rows = np.arange(m)
        bases[rows, kb + pstar] = 1
        if cg_context:
            bases[rows, kb + pstar + 1] = 2
        seqs[st : st + m] = bases
        signal[st : st + m, 0] = rng.standard_normal((m, L), dtype=np.float32)
    labels = rng.integers(0, num_out, n).astype(np.int64)
    return dict(signal=signal, sequence=seqs, sequence_to_signal_mapping=maps, sequence_lengths=lens,
                labels=labels, kmer_context_bases=(kb, ka), chunk_len=L)


def synth_chunks_config(name, n, seed=20240926, shard=0):
    cc, kcb, msl, num_out, cg = CONFIGS[name]
    return synth_chunks(n, sum(cc), msl, kcb, num_out, cg, seed, shard)


def synth_read(n_bases=5000, dwell_lo=5, dwell_hi=15, seed=20240926, idx=0):
    """SURVEY §8(d) synthetic read: bases ~ U[0,3], dwell ~ U[5,15], dacs ~ U[300,700]."""
    rng = np.random.default_rng(seed * 7919 + idx)
    int_seq = rng.integers(0, 4, n_bases).astype(np.int64)
    dwell = rng.integers(dwell_lo, dwell_hi + 1, n_bases)
    s2s = np.concatenate([[0], np.cumsum(dwell)]).astype(np.int64)
    dacs = rng.integers(300, 701, s2s[-1]).astype(np.int16)
    return dict(dacs=dacs, seq_to_sig_map=s2s, int_seq=int_seq, shift=500.0, scale=80.0)


def synth_state(arch="conv_lstm", size=64, kmer_len=9, num_out=2, seed=0, amplify=True):
    """Random weights with the reference architectures' tensor names/shapes
    (models/ConvLSTM_w_ref.py:11-37, models/Conv_w_ref.py:11-42): uniform(-1/sqrt(fan_in), ..)
    like torch's default init, BatchNorm running stats randomised (mean~N(0,1), var~U(0.5,2),
    gamma~N(1,0.2), beta~N(0,0.2)) so that folding matters.  `amplify` (default): conv weights x 2.45, LSTM weights x 2.5,
    forget-gate bias + 2, fc x 12 - a network that carries input variation (and rounding) to the logits, for tests that
    must not be blind to early layers; False: torch's default bounds everywhere (the scale of a reference-initialised
    network, e.g. the golden models)."""
    rng = np.random.default_rng(1000 + seed)
    st = {}
    ca, la, fa = (2.45, 2.5, 12.0) if amplify else (1.0, 1.0, 1.0)

    def conv(name, bn, ic, oc, k):
        b = 1.0 / np.sqrt(ic * k)
        # He-scale weights (x2.45 torch's default bound) so that input variation survives to the
        # logits: with the default bound the random BN offsets swamp the signal and every chunk
        # gets nearly the same logits, which would make parity tests blind to early-layer bugs
        st[f"{name}.weight"] = rng.uniform(-ca * b, ca * b, (oc, ic, k)).astype(np.float32)
        st[f"{name}.bias"] = rng.uniform(-b, b, oc).astype(np.float32)
        st[f"{bn}.weight"] = (1.0 + 0.2 * rng.standard_normal(oc)).astype(np.float32)
        st[f"{bn}.bias"] = (0.2 * rng.standard_normal(oc)).astype(np.float32)
        st[f"{bn}.running_mean"] = rng.standard_normal(oc).astype(np.float32)
        st[f"{bn}.running_var"] = rng.uniform(0.5, 2.0, oc).astype(np.float32)

    ec = 4 * kmer_len
    if arch == "conv_lstm":
        conv("sig_conv1", "sig_bn1", 1, 4, 5); conv("sig_conv2", "sig_bn2", 4, 16, 5)
        conv("sig_conv3", "sig_bn3", 16, size, 9)
        conv("seq_conv1", "seq_bn1", ec, 16, 5); conv("seq_conv2", "seq_bn2", 16, size, 13)
        conv("merge_conv1", "merge_bn", 2 * size, size, 5)
        b = 1.0 / np.sqrt(size)
        for l in ("lstm1", "lstm2"):
            st[f"{l}.weight_ih_l0"] = rng.uniform(-la * b, la * b, (4 * size, size)).astype(np.float32)
            st[f"{l}.weight_hh_l0"] = rng.uniform(-la * b, la * b, (4 * size, size)).astype(np.float32)
            st[f"{l}.bias_ih_l0"] = rng.uniform(-b, b, 4 * size).astype(np.float32)
            if amplify:
                st[f"{l}.bias_ih_l0"][size : 2 * size] += 2.0  # forget-gate bias: longer memory, as in trained LSTMs
            st[f"{l}.bias_hh_l0"] = rng.uniform(-b, b, 4 * size).astype(np.float32)
        fin = size
    else:
        conv("sig_conv1", "sig_bn1", 1, 4, 11); conv("sig_conv2", "sig_bn2", 4, 16, 11)
        conv("sig_conv3", "sig_bn3", 16, size, 9)
        conv("seq_conv1", "seq_bn1", ec, 16, 11); conv("seq_conv2", "seq_bn2", 16, 32, 11)
        conv("seq_conv3", "seq_bn3", 32, size, 9)
        conv("merge_conv1", "merge_bn1", 2 * size, size, 5); conv("merge_conv2", "merge_bn2", size, size, 5)
        conv("merge_conv3", "merge_bn3", size, size, 3); conv("merge_conv4", "merge_bn4", size, size, 3)
        fin = size * 3
    b = ((fa if arch == "conv_lstm" else 1.5) if amplify else 1.0) / np.sqrt(fin)  # (amplified: logits within a few units)
    st["fc.weight"] = rng.uniform(-b, b, (num_out, fin)).astype(np.float32)
    st["fc.bias"] = rng.uniform(-b, b, num_out).astype(np.float32)
    return st
