"""Host front-end alone on the box's cores: NativeBam.read_batch over a BAM written from the first N records of the configs[1] batch,
threads x {dense, sparse SEQ}.  Usage: python tools/reader_scaling.py [n_records]   (needs the GPU only to generate the batch)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                             # noqa: E402
from svim_amd import devsynth, harness                   # noqa: E402
from svim_amd.bamio import NativeBam                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
b, genome, meta = devsynth.make_batch(n_reads=max(n, 1000), n50=20000, contig_len=max(3_000_000, 250 * n), seed=2, device=dev)
hb = b.slice_records(0, min(n, b.n_rec))
path = "/tmp/reader_scaling.bam"
t0 = time.perf_counter()
nrec, raw = harness.write_bam_from_batch(path, hb, ["chr1"], [int(genome.numel())])
size = os.path.getsize(path)
print("%d records, BAM %.0f MB (%.0f MB inflated), written in %.1f s; host cores %d" % (nrec, size / 1e6, raw / 1e6, time.perf_counter() - t0, os.cpu_count()))
for sparse in (0, 40):
    for threads in (1, 8, 32, 64, 128, 256):
        if threads > (os.cpu_count() or 1):
            continue
        best, first = 1e9, None
        nb = NativeBam(path, threads=threads)
        if sparse:
            nb.set_seq_filter(sparse)
        for it in range(4):                                  # pass 0 is cold (first-touch allocation), the others reuse every buffer
            if it:
                nb.rewind()
            t = time.perf_counter()
            tot = 0
            while True:
                bb, k = nb.read_batch(10000, 20, "coordinate")
                if k == 0:
                    break
                tot += k
            dt = time.perf_counter() - t
            if it == 0:
                first = dt
            else:
                best = min(best, dt)
        nb.close()
        print("seq %s threads %3d: %.3f s (first pass %.3f s)  %.2f M records/s  %.0f MB/s of BAM  %.2f GB/s inflated" % (
            "sparse" if sparse else "dense ", threads, best, first, tot / best / 1e6, size / best / 1e6, raw / best / 1e9))
os.remove(path)
