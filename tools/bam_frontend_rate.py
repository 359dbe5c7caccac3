"""Rate of the C++ BAM front-end (BGZF inflate + record decode + SA / grouping -> SoA batch) on the host cores, no GPU needed.
Writes a synthetic coordinate-sorted BAM of ONT-like reads (own writer, svim_amd/records.py:write_bam), then times
NativeBam.read_batch for 1..N threads.  Usage: python tools/bam_frontend_rate.py [n_reads] [read_len]"""
import os
import random
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svim_amd import records, synth          # noqa: E402
from svim_amd.bamio import NativeBam         # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
rng = random.Random(3)
references, lengths = ["chr1"], [50_000_000]
recs = []
for i in range(n_reads):
    qlen = max(500, int(rng.gauss(read_len, read_len / 3)))
    core = synth.noisy_core(rng, qlen)                     # [(op, len)]: M runs with short indels, ~1 op / 9 bp
    a = records.AlignedSegment()
    a.query_name = "r%06d" % i
    a.flag = 16 if rng.random() < 0.5 else 0
    a.reference_id = 0
    a.reference_start = rng.randint(0, lengths[0] - 3 * qlen)
    a.mapping_quality = 60
    a.cigartuples = core
    a.query_sequence = synth.random_seq(rng, sum(l for op, l in core if op in (0, 1, 4, 7, 8)))
    recs.append(a)
recs.sort(key=lambda r: r.reference_start)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "t.bam")
    records.write_bam(path, references, lengths, recs)
    size = os.path.getsize(path)
    n_ops = sum(len(r.cigartuples) for r in recs)
    bases = sum(len(r.query_sequence) for r in recs)
    print("%d records, %.1f M CIGAR ops, %.1f Mb of sequence, BAM %.1f MB" % (n_reads, n_ops / 1e6, bases / 1e6, size / 1e6))
    for threads in (1, 2, 4, 8):
        best = 1e9
        for _ in range(3):
            nb = NativeBam(path, threads=threads)
            t = time.perf_counter()
            b = nb.read_batch(1 << 30, 20, mode="coordinate")
            dt = time.perf_counter() - t
            nb.close()
            best = min(best, dt)
        print("threads %d: %.3f s  %.2f M records/s  %.0f MB/s of BAM  %.1f M CIGAR ops/s" %
              (threads, best, n_reads / best / 1e6, size / best / 1e6, n_ops / best / 1e6))
