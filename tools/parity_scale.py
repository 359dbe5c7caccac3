"""GPU vs oracle on a device-generated batch (bench-like data at reduced size): full table comparison."""
import sys, time, types
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svim_amd import _abi, _lib, devsynth
from oracle import oracle as om
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
contig = int(sys.argv[2]) if len(sys.argv) > 2 else 15_000_000
pmd = int(sys.argv[3]) if len(sys.argv) > 3 else 1000              # options.partition_max_distance (C5 sweeps 1000 .. 100000)
all_bnds = len(sys.argv) > 4 and sys.argv[4] == "all_bnds"
o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                          partition_max_distance=pmd, position_distance_normalizer=900, edit_distance_normalizer=1.0, cluster_max_distance=0.5, all_bnds=all_bnds)
p = _abi.Params.from_options(o)
b, genome, meta = devsynth.make_batch(n_reads=n_reads, contig_len=contig, seed=2, device="cuda:0")
eng = _lib.Engine(0)
eng.set_genome(torch.tensor([0, genome.numel()], dtype=torch.int64, device="cuda:0"), genome, on_device=True)
sig, bnd = eng.collect(b.struct(), p)
ct = eng.cluster(p, np.zeros(1, np.int32), source=0)
print("gpu:", sig.n, ct.n, eng.stats()["n_edit_pairs"])
orc = om.Oracle()
g = genome.cpu().numpy()
orc.set_genome(np.array([0, g.size], dtype=np.int64), g)
hb = b.slice_records(0, b.n_rec)
t = time.time()
osig, obnd = orc.collect(hb, p)
oc = orc.cluster(p, np.zeros(1, np.int32), source=0)
print("oracle:", osig.n, oc.n, "%.1fs" % (time.time() - t))
d1 = sig.first_difference(osig)
d2 = ct.first_difference(oc, rtol=1e-12)
print("sig diff:", d1, "| cluster diff:", d2, "| side list diff:", bnd.first_difference(obnd) if all_bnds else "n/a")
if d2:
    # locate the first differing cluster by partition
    k = 0
    while k < min(ct.n, oc.n) and ct.size[k] == oc.size[k] and ct.start[k] == oc.start[k] and ct.end[k] == oc.end[k] and ct.type[k] == oc.type[k]:
        k += 1
    print("first differing cluster", k, "gpu", ct.type[k], ct.start[k], ct.end[k], ct.size[k], "oracle", oc.type[k], oc.start[k], oc.end[k], oc.size[k])
    lo = max(0, k - 1)
    for kk in range(lo, min(lo + 4, ct.n)):
        print(" gpu", kk, ct.type[kk], ct.start[kk], ct.end[kk], ct.size[kk], ct.part_index[kk], " | orc", oc.type[kk], oc.start[kk], oc.end[kk], oc.size[kk], oc.part_index[kk])
    # dump the INS signatures of that partition for offline analysis
    import pickle
    mem = oc.members[oc.member_off[k]:oc.member_off[k + 1]]
    s0 = int(osig.start[mem[0]])
    sel = [i for i in range(osig.n) if osig.type[i] == osig.type[mem[0]] and abs(int(osig.start[i]) - s0) < 3000]
    out = dict(sel=sel, start=[int(osig.start[i]) for i in sel], end=[int(osig.end[i]) for i in sel], read=[int(osig.read_id[i]) for i in sel],
               seq=[osig.sequence(i) for i in sel], ref=_abi.decode_bases(g[max(0, s0 - 5000):s0 + 8000]), ref_off=max(0, s0 - 5000))
    pickle.dump(out, open("gpurun_out/parity_fail.pkl", "wb"))
