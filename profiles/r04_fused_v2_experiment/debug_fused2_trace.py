"""Off-line half of the RMR_FUSED2_DBG=512 experiment (identical chunks): where do the wrong SIG1 words come from?
Reads the file written through RMR_FUSED2_DBGFILE: [16 ints layout][counters 128][log 4096][LDS image]."""
import sys

import numpy as np

raw = open(sys.argv[1], "rb").read()
hdr = np.frombuffer(raw[:64], dtype=np.int32)
names = ["lds_bytes", "o_sig", "o_seq", "o_map", "o_len", "o_tab", "o_bias", "o_col3", "o_col4", "o_sig1", "o_sig2", "o_seq1", "o_oh", "o_cat", "oh_plane", "cat_plane"]
L = dict(zip(names, hdr.tolist()))
print(L)
body = np.frombuffer(raw[64:], dtype=np.uint64)
counters, log, img64 = body[:128], body[128:4224], body[4224:]
img = img64.view(np.uint32)[: L["lds_bytes"] // 4]
nlog = int(min(counters[127], 4096))
print("logged words", int(counters[127]))
sig1 = img[L["o_sig1"] // 4: L["o_sig2"] // 4]
regions = [(n, L[n]) for n in ("o_sig", "o_seq", "o_map", "o_len", "o_tab", "o_bias", "o_col3", "o_col4", "o_sig1", "o_sig2", "o_seq1", "o_oh", "o_cat")]


def where(byte_off):
    name = [n for n, o in regions if o <= byte_off][-1]
    return f"{name}+{byte_off - L[name]}"


bad = []
for e in log[:nlog]:
    i, v = int(e >> np.uint64(32)), int(e & np.uint64(0xFFFFFFFF))
    if i < sig1.size and int(sig1[i]) != v:
        bad.append((i, v, int(sig1[i])))
print("words of the logged slices that differ from the image's SIG1:", len(bad))
seen = 0
for i, v, exp in sorted(set(bad))[:60]:
    hits = np.nonzero(img == v)[0]
    rows = f"SIG1 row {i // 2} half {i % 2} (byte {L['o_sig1'] + 4 * i})"
    print(f"  {rows}: got {v:08x} expected {exp:08x}; the wrong word occurs in the image at: {[where(4 * int(h)) for h in hits[:6]]}")
