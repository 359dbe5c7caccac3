"""Repeat COLLECT+CLUSTER in one context and compare every pass with the first (hunts order-dependent bugs: the INS pair
work list is built with atomics, so the grouping of pairs into waves differs from pass to pass)."""
import os, sys, types, pickle
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from svim_amd import _abi, _lib, devsynth
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
contig = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                          partition_max_distance=1000, position_distance_normalizer=900, edit_distance_normalizer=1.0, cluster_max_distance=0.5, all_bnds=False)
p = _abi.Params.from_options(o)
b, genome, meta = devsynth.make_batch(n_reads=n_reads, contig_len=contig, seed=2, device="cuda:0")
eng = _lib.Engine(0)
eng.set_genome(torch.tensor([0, genome.numel()], dtype=torch.int64, device="cuda:0"), genome, on_device=True)
first = None
redo_collect = os.environ.get("REDO_COLLECT", "1") == "1"
eng.collect(b.struct(), p, fetch=False)
nbad = 0
for r in range(reps):
    if redo_collect:
        eng.collect(b.struct(), p, fetch=False)
    ct = eng.cluster(p, np.zeros(1, np.int32), source=0)
    if first is None:
        first = ct
        print("pass 0 clusters", ct.n)
        continue
    d = ct.first_difference(first)
    if d or r % 10 == 0:
        print("pass", r, "clusters", ct.n, "diff:", d)
    if d:
        nbad += 1
        a, f = ct, first
        k = 0
        while k < min(a.n, f.n) and a.size[k] == f.size[k] and a.start[k] == f.start[k] and a.end[k] == f.end[k] and a.type[k] == f.type[k]:
            k += 1
        for kk in range(max(0, k - 1), min(k + 3, a.n)):
            print("  now", kk, a.type[kk], a.start[kk], a.end[kk], a.size[kk], a.part_index[kk], "| first", f.type[kk], f.start[kk], f.end[kk], f.size[kk], f.part_index[kk])
        sig = eng.fetch_signatures(0)
        part = int(f.part_index[k])
        ks = [i for i in range(f.n) if f.part_index[i] == part]
        mem = np.concatenate([f.members[f.member_off[i]:f.member_off[i + 1]] for i in ks])
        g = genome.cpu().numpy()
        s0 = int(sig.start[mem[0]])
        sel = [int(i) for i in np.nonzero((sig.type == sig.type[mem[0]]) & (np.abs(sig.start.astype(np.int64) - s0) < 3000))[0]]
        out = dict(sel=sel, start=[int(sig.start[i]) for i in sel], end=[int(sig.end[i]) for i in sel], read=[int(sig.read_id[i]) for i in sel],
                   seq=[sig.sequence(i) for i in sel], ref=_abi.decode_bases(g[max(0, s0 - 5000):s0 + 8000]), ref_off=max(0, s0 - 5000))
        pickle.dump(out, open("gpurun_out/repeat_fail.pkl", "wb"))
        print("dumped", len(sel), "signatures around", s0)
        if nbad >= 3:
            break
print("mismatching passes:", nbad, "of", reps)
