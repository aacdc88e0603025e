"""Latency of call_read_mods (one 5 kb read per call -> rmr_call_read) with the fp32 and the bf16 model: median / best of 5 passes
over 128 reads after 8 warm-up calls (what bench.py's reads leg reports as single_read_api_us_per_read)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remora_amd import synth  # noqa: E402
from remora_amd.data_chunks import RemoraRead  # noqa: E402
from remora_amd.inference import call_read_mods  # noqa: E402
from remora_amd.model_util import model_from_state  # noqa: E402

md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"], can_base="C",
          base_start_justify=False, offset=0, sig_map_refiner=None)
rs = []
for i in range(128):
    r = synth.synth_read(5000, idx=i)
    rs.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"], int_seq=r["int_seq"], read_id=f"s{i}"))
for dt in ("fp32", "bf16"):
    model = model_from_state(synth.synth_state(seed=0), md, device=0, dtype=dt)
    for r in rs[:8]:
        call_read_mods(r, model, md)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for r in rs:
            call_read_mods(r, model, md)
        ts.append((time.perf_counter() - t0) / 128)
    ts.sort()
    print(f"{dt}: {ts[2] * 1e6:.1f} us per read (median of 5 passes; best {ts[0] * 1e6:.1f}) = {1 / ts[2]:.0f} reads/s")
