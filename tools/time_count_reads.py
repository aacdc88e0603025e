"""Wall time of prepare's counting pass (prepare_train_data.count_reads) on the replicated test alignments, by the number of
inflate threads of the native BAM reader.   python tools/time_count_reads.py [REP=12000]"""
import os
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    from remora_amd import prepare_train_data as p

    t = time.perf_counter()
    got = p.count_reads(sys.argv[2], sys.argv[3])
    print(f"{os.environ.get('RMR_BAM_INFLATE_THREADS', 'default')} inflate threads: {got} in {time.perf_counter() - t:.2f} s", flush=True)
    sys.exit(0)
from remora_amd import io as rio  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
data = os.path.join(ROOT, "tests", "golden", "data")
pod5, bam = os.path.join(data, "mod_reads.pod5"), os.path.join(data, "mod_mappings.bam")
big = os.path.join(tempfile.mkdtemp(), "big.bam")
recs = list(rio.iter_bam_records(bam, want_ref=False))
with rio.BamWriter(big, rio.read_bam_header_bytes(bam), level=1) as w:
    for _ in range(REP):
        for r in recs:
            raw = bytes(r.raw)
            w.write(struct.pack("<i", len(raw)) + raw)
print(f"{REP * len(recs)} records, {os.path.getsize(big) / 1e6:.0f} MB", flush=True)
for th in ("4", "8", "12", "16", "8"):
    subprocess.run([sys.executable, __file__, "--child", pod5, big], env=dict(os.environ, RMR_BAM_INFLATE_THREADS=th))
