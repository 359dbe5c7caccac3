"""GPU BGZF inflate (svx_inflater) on the configs[1]-like sample BAM of the end-to-end block: kernel GB/s of inflated output, against zlib on the
host's cores.  Usage: python tools/bgzf_inflate_rate.py [n_records]"""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                        # noqa: E402
import torch                                              # noqa: E402
from svim_amd import devsynth, harness                    # noqa: E402
from svim_amd._lib import Inflater, bgzf_blocks           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
b, genome, meta = devsynth.make_batch(n_reads=max(n, 1000), n50=20000, contig_len=max(3_000_000, 250 * n), seed=2, device="cuda:0")
hb = b.slice_records(0, min(n, b.n_rec))
path = "/tmp/bgzf_rate.bam"
nrec, raw = harness.write_bam_from_batch(path, hb, ["chr1"], [int(genome.numel())])
blocks = bgzf_blocks(path)
comp = sum(len(p) for p, _ in blocks)
out_bytes = sum(s for _, s in blocks)
print("%d records, %d BGZF blocks, %.1f MB compressed -> %.1f MB (ratio %.2f)" % (nrec, len(blocks), comp / 1e6, out_bytes / 1e6, out_bytes / max(1, comp)))
t0 = time.perf_counter()
ref = [zlib.decompress(p, -15) if s else b"" for p, s in blocks[:2000]]
dt = time.perf_counter() - t0
print("zlib, one host thread (first 2000 blocks): %.2f GB/s of inflated output" % (sum(len(r) for r in ref) / dt / 1e9))
f = Inflater(0)
best = None
for chunk in (len(blocks), 8192, 2048):
    for rep in range(3):
        t0 = time.perf_counter()
        ms = 0.0
        for lo in range(0, len(blocks), chunk):
            got = f.inflate(blocks[lo:lo + chunk])
            ms += f.kernel_ms
        wall = time.perf_counter() - t0
    print("GPU, %6d blocks per launch: kernel %.2f ms = %.1f GB/s inflated (%.1f GB/s of compressed input); python wall incl. packing + H2D + D2H %.2f s" % (
        chunk, ms, out_bytes / ms / 1e6, comp / ms / 1e6, wall))
got = f.inflate(blocks[:2000])
assert got.tobytes() == b"".join(ref), "GPU inflate differs from zlib"
print("first 2000 blocks identical to zlib")
# the same records with base qualities (random Phred values): literal-heavy blocks, the demanding case for a Huffman decoder
hq = b.slice_records(0, min(n // 2, b.n_rec))
nrec, raw = harness.write_bam_from_batch("/tmp/bgzf_rate_q.bam", hq, ["chr1"], [int(genome.numel())], qual_seed=7)
blocks = bgzf_blocks("/tmp/bgzf_rate_q.bam")
comp = sum(len(p) for p, _ in blocks)
out_bytes = sum(s for _, s in blocks)
print("with base qualities: %d records, %d BGZF blocks, %.1f MB compressed -> %.1f MB (ratio %.2f)" % (nrec, len(blocks), comp / 1e6, out_bytes / 1e6, out_bytes / max(1, comp)))
t0 = time.perf_counter()
ref = [zlib.decompress(p, -15) if s else b"" for p, s in blocks[:2000]]
dt = time.perf_counter() - t0
print("zlib, one host thread (first 2000 blocks): %.2f GB/s of inflated output" % (sum(len(r) for r in ref) / dt / 1e9))
for rep in range(3):
    got = f.inflate(blocks)
    ms = f.kernel_ms
print("GPU, %6d blocks per launch: kernel %.2f ms = %.1f GB/s inflated (%.1f GB/s of compressed input)" % (len(blocks), ms, out_bytes / ms / 1e6, comp / ms / 1e6))
got = f.inflate(blocks[:2000])
assert got.tobytes() == b"".join(ref), "GPU inflate differs from zlib"
print("first 2000 blocks identical to zlib")
f.close()
