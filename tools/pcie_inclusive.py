"""PCIe-inclusive rate of one COLLECT+CLUSTER pass: the same synthetic batch handed over as HOST arrays (svx_collect uploads them)
versus resident in HBM.  Usage: python tools/pcie_inclusive.py [n_reads] [contig_len]"""
import sys
import time
import types

sys.path.insert(0, "/root/repo")
import numpy as np      # noqa: E402
import torch            # noqa: E402
from svim_amd import _abi, _lib, devsynth   # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
contig = int(sys.argv[2]) if len(sys.argv) > 2 else 62_500_000
o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                          partition_max_distance=1000, position_distance_normalizer=900, edit_distance_normalizer=1.0,
                          cluster_max_distance=0.5, all_bnds=False)
p = _abi.Params.from_options(o)
b, genome, meta = devsynth.make_batch(n_reads=n_reads, contig_len=contig, seed=2, device="cuda:0")
eng = _lib.Engine(0)
eng.set_genome(torch.tensor([0, genome.numel()], dtype=torch.int64, device="cuda:0"), genome, on_device=True)
hb = b.slice_records(0, b.n_rec)                 # host copy of the whole batch (numpy arrays)
rank = np.zeros(1, np.int32)


def run(batch_struct):
    torch.cuda.synchronize()
    t = time.perf_counter()
    eng.collect(batch_struct, p, fetch=False)
    eng.cluster(p, rank, source=0, fetch=False)
    torch.cuda.synchronize()
    return time.perf_counter() - t


for _ in range(2):
    run(b.struct()); run(hb)
dev_t = min(run(b.struct()) for _ in range(3))
host_t = min(run(hb) for _ in range(3))
used = eng.stats()["n_rec_used"]
nbytes = sum(v.nbytes for v in hb.arrays.values())
print("records used %d, host batch %.2f GB" % (used, nbytes / 1e9))
print("resident in HBM : %.1f ms/pass  %.2f M records/s" % (dev_t * 1e3, used / dev_t / 1e6))
print("host arrays in  : %.1f ms/pass  %.2f M records/s  (upload %.1f ms = %.1f GB/s effective)" %
      (host_t * 1e3, used / host_t / 1e6, (host_t - dev_t) * 1e3, nbytes / 1e9 / max(1e-9, host_t - dev_t)))
