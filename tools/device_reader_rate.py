"""The device-resident BAM reader (svx_bam_set_device_decode: inflate + record discovery + decode on the GPU) alone: records/s, GB/s of inflated
data, and where the time of a pass goes (SVX_BAM_TIMING stage times from the library); next to the host reader with GPU-assisted inflate and the
host-only reader.  The clock of a pass starts BEFORE rewind().  Usage: python tools/device_reader_rate.py [n_records] [chunk_MB ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                              # noqa: E402
from svim_amd import devsynth, harness                    # noqa: E402
from svim_amd.bamio import NativeBam                      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 180000
chunks = [int(x) for x in sys.argv[2:]] or [2048]
b, genome, meta = devsynth.make_batch(n_reads=max(n, 1000), n50=20000, contig_len=max(3_000_000, 250 * n), seed=2, device="cuda:0")
hb = b.slice_records(0, min(n, b.n_rec))
path = "/tmp/device_reader.bam"
nrec, raw = harness.write_bam_from_batch(path, hb, ["chr1"], [int(genome.numel())])
size = os.path.getsize(path)
print("%d records, BAM %.0f MB (%.0f MB inflated)" % (nrec, size / 1e6, raw / 1e6))
del b, hb
torch.cuda.empty_cache()
os.environ["SVX_BAM_TIMING"] = "1"


def passes(nb, k=3):
    best = 1e9
    for it in range(k):
        t = time.perf_counter()
        if it:
            nb.rewind()
        tot = 0
        while True:
            bb, m = nb.read_batch(30000, 20, "coordinate")
            if m == 0:
                break
            tot += m
        dt = time.perf_counter() - t
        print("   pass %d: %.3f s" % (it, dt), flush=True)
        if it:
            best = min(best, dt)
    return tot, best


variants = [("GPU + host cores, staged input", {}), ("GPU only (default input path)", {"SVX_BAM_DEV_CPU": "0"}), ("GPU only, sub-batches of 12288", {"SVX_BAM_DEV_CPU": "0"}), ("GPU only, sub-batches of 4096", {"SVX_BAM_DEV_CPU": "0", "SVX_BAM_DEV_SUB": "4096"}),
            ("GPU only, sub-batches of 40000", {"SVX_BAM_DEV_CPU": "0", "SVX_BAM_DEV_SUB": "40000"}), ("GPU + 8 host threads", {"SVX_BAM_DEV_CPU": "8"}),
            ("GPU + 6 host threads, sub-batches of 24576", {"SVX_BAM_DEV_CPU": "6", "SVX_BAM_DEV_SUB": "24576"})]
if os.environ.get("SVX_READER_ONE"):                  # (profiling runs: the GPU-only reader with mid-size sub-batches, nothing else)
    variants = [("GPU only, sub-batches of 12288", {"SVX_BAM_DEV_CPU": "0", "SVX_BAM_DEV_SUB": "12288"})]
for mb in chunks:
    for label, env in variants if mb == chunks[0] else variants[:1]:
        os.environ["SVX_BAM_DEV_CHUNK_MB"] = str(mb)
        for k in ("SVX_BAM_DEV_CPU", "SVX_BAM_DEV_SUB"):
            os.environ.pop(k, None)
        os.environ.update(env)
        nb = NativeBam(path)
        nb.set_device_decode(0)
        tot, t = passes(nb)
        print("device reader, %4d MB chunks, %s: %.3f s  %.2f M records/s  %.1f GB/s inflated  %.1f GB/s of BAM  %r" % (
            mb, label, t, tot / t / 1e6, raw / t / 1e9, size / t / 1e9, nb.gpu_inflate_stats()), flush=True)
        nb.close()                                        # (prints the stage times of all passes to stderr)
for k in ("SVX_BAM_DEV_CPU", "SVX_BAM_DEV_SUB"):
    os.environ.pop(k, None)
if os.environ.get("SVX_READER_ONE"):
    sys.exit(0)
nb = NativeBam(path)
nb.set_seq_filter(40)
nb.set_gpu_inflate(0)
tot, t = passes(nb)
print("host reader + GPU-assisted inflate: %.3f s  %.2f M records/s  %.1f GB/s inflated" % (t, tot / t / 1e6, raw / t / 1e9), flush=True)
nb.close()
