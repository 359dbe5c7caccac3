"""GENOTYPE at configs[1] scale: every DEL / INS cluster of the synthetic batch as a candidate, the batch's own records as the
alignment index (reference_end approximated by pos + read length: timing and GPU-vs-oracle parity only, not a reference comparison)."""
import sys
import time
import types

sys.path.insert(0, "/root/repo")
import numpy as np      # noqa: E402
import torch            # noqa: E402
from svim_amd import _abi, _lib, devsynth   # noqa: E402
from oracle import oracle as om             # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
contig = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000_000
o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                          partition_max_distance=1000, position_distance_normalizer=900, edit_distance_normalizer=1.0,
                          cluster_max_distance=0.5, all_bnds=False)
p = _abi.Params.from_options(o)
b, genome, meta = devsynth.make_batch(n_reads=n_reads, contig_len=contig, seed=2, device="cuda:0")
eng = _lib.Engine(0)
eng.set_genome(torch.tensor([0, genome.numel()], dtype=torch.int64, device="cuda:0"), genome, on_device=True)
sig, _ = eng.collect(b.struct(), p)
ct = eng.cluster(p, np.zeros(1, np.int32), source=0)


class Index(object):
    pass


t = b.t
ix = Index()
ix.n, ix.n_contig = int(b.n_rec), 1
ix.contig_first = np.array([0, b.n_rec], dtype=np.int64)
ix.contig_len = np.array([contig], dtype=np.int64)
ix.pos = t["pos"].cpu().numpy().astype(np.int32)
ix.end = (ix.pos.astype(np.int64) + t["lseq"].cpu().numpy()).clip(max=contig).astype(np.int32)
ix.flag = t["flag"].cpu().numpy().astype(np.uint16)
ix.mapq = t["mapq"].cpu().numpy().astype(np.uint8)
ix.name_id = t["read_id"].cpu().numpy().astype(np.int32)


def view():
    v = _abi.AlnIndex()
    v.n, v.n_contig = ix.n, ix.n_contig
    v.contig_first, v.contig_len, v.pos, v.end, v.flag, v.mapq, v.name_id = [_abi.ptr(a) for a in
                                                                            (ix.contig_first, ix.contig_len, ix.pos, ix.end, ix.flag, ix.mapq, ix.name_id)]
    return v


ix.view = view
t0 = time.perf_counter()
eng.set_alignment_index(ix)
t_index = time.perf_counter() - t0
orc = om.Oracle()
orc.set_alignment_index(ix)
for typ, mode in ((0, 0), (1, 1)):                     # DEL clusters, INS clusters
    sel = np.nonzero(ct.type[:ct.n] == typ)[0]
    tid = np.zeros(sel.size, dtype=np.int32)
    start = ct.start[sel].astype(np.int32)
    end = start.copy() if mode else ct.end[sel].astype(np.int32)
    moff = [0]
    mnames = []
    for k in sel:
        ids = np.unique(sig.read_id[ct.members[ct.member_off[k]:ct.member_off[k + 1]]])
        mnames.append(ids)
        moff.append(moff[-1] + ids.size)
    mnames = np.concatenate(mnames).astype(np.int32)
    eng.genotype(mode, tid, start, end, moff, mnames, 20)
    t0 = time.perf_counter()
    ref = eng.genotype(mode, tid, start, end, moff, mnames, 20)
    dt = time.perf_counter() - t0
    sub = slice(0, 300)
    t0 = time.perf_counter()
    oref = orc.genotype(mode, tid[sub], start[sub], end[sub], np.asarray(moff[:301]), mnames[:moff[300]], 20)
    dto = time.perf_counter() - t0
    print("%s: %d candidates, %.2f ms on the GPU incl. transfers (%.2f M candidates/s), mean ref reads %.1f; oracle %.1f us/candidate; first 300 identical: %s"
          % ("INS" if mode else "DEL", sel.size, dt * 1e3, sel.size / dt / 1e6, ref.mean(), dto / 300 * 1e6, bool((ref[sub] == oref).all())))
print("index upload + running-maximum pass: %.1f ms for %d records" % (t_index * 1e3, ix.n))
