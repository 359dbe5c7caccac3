"""The BAM reader alone with BGZF inflate shared between the GPU and the host's cores (svx_bam_set_gpu_inflate) against the host-only reader:
records/s, GB/s of inflated data, how many blocks either side took, and a check that both deliver the same arrays.
Usage: python tools/reader_gpu_inflate.py [n_records]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                        # noqa: E402
import torch                                              # noqa: E402
from svim_amd import devsynth, harness                    # noqa: E402
from svim_amd.bamio import NativeBam                      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 240000
dev = "cuda:0"
b, genome, meta = devsynth.make_batch(n_reads=max(n, 1000), n50=20000, contig_len=max(3_000_000, 250 * n), seed=2, device=dev)
hb = b.slice_records(0, min(n, b.n_rec))
path = "/tmp/reader_gpu.bam"
t0 = time.perf_counter()
nrec, raw = harness.write_bam_from_batch(path, hb, ["chr1"], [int(genome.numel())])
size = os.path.getsize(path)
print("%d records, BAM %.0f MB (%.0f MB inflated), written in %.1f s" % (nrec, size / 1e6, raw / 1e6, time.perf_counter() - t0))
del b, hb
torch.cuda.empty_cache()


def run(gpu, threads=0, sub=None, passes=3, checksum=False):
    if sub:
        os.environ["SVX_BAM_GPU_SUB"] = str(sub)
    nb = NativeBam(path, threads=threads)
    nb.set_seq_filter(40)
    if gpu:
        nb.set_gpu_inflate(0)
    best, sums = 1e9, None
    for it in range(passes):
        t = time.perf_counter()
        if it:
            nb.rewind()                             # inside the clock: svx_bam_rewind inflates the first chunk
        tot, acc = 0, 0
        while True:
            bb, k = nb.read_batch(100000, 20, "coordinate")
            if k == 0:
                break
            tot += k
            if checksum and it == 0:
                arr = nb.batch_arrays(bb) if hasattr(nb, "batch_arrays") else None
                if arr is not None:
                    acc += int(arr["pos"].astype(np.int64).sum()) + int(arr["cigar"][:int(arr["cigar_off"][k])].astype(np.int64).sum()) + int(arr["flag"].astype(np.int64).sum())
        dt = time.perf_counter() - t
        if it:
            best = min(best, dt)
        if checksum and it == 0:
            sums = acc
    st = nb.gpu_inflate_stats() if gpu else {}
    nb.close()
    return tot, best, st, sums


tot, t_cpu, _, s_cpu = run(False, checksum=True)
print("host only              : %.3f s  %.2f M records/s  %.2f GB/s inflated" % (t_cpu, tot / t_cpu / 1e6, raw / t_cpu / 1e9))
for sub in (8192, 16384, 32768):
    tot2, t_gpu, st, s_gpu = run(True, sub=sub, checksum=True)
    share = st["gpu_blocks"] / max(1, st["gpu_blocks"] + st["cpu_blocks"])
    print("GPU + host, sub %5d   : %.3f s  %.2f M records/s  %.2f GB/s inflated   GPU took %.0f %% of the blocks (kernels %.0f ms over all passes)  same arrays: %s" % (
        sub, t_gpu, tot2 / t_gpu / 1e6, raw / t_gpu / 1e9, 100 * share, st["gpu_kernel_ms"], s_gpu == s_cpu and tot2 == tot))
os.remove(path)
