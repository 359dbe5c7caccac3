"""Latency of one COLLECT + CLUSTER call on small batches (the per-read entry points of the drop-in launch the whole pipeline for one record):
wall time per call for batches of 1 ... 100 000 reads.  Usage: python tools/small_batch_latency.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                              # noqa: E402
import numpy as np                                        # noqa: E402
from svim_amd import _abi, _lib, devsynth                 # noqa: E402

o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                          partition_max_distance=1000, position_distance_normalizer=900, edit_distance_normalizer=1.0, cluster_max_distance=0.5, all_bnds=False)
p = _abi.Params.from_options(o)
eng = _lib.Engine(0)
for n in (1, 10, 100, 1000, 10000, 100000):
    b, genome, meta = devsynth.make_batch(n_reads=n, contig_len=max(2_000_000, 250 * n), seed=2, device="cuda:0")
    g_off = torch.tensor([0, genome.numel()], dtype=torch.int64, device="cuda:0")
    eng.set_genome(g_off, genome, on_device=True)
    rank = b.t["contig_rank"].cpu().numpy().astype(np.int32)
    bs = b.struct()
    for _ in range(3):
        eng.collect(bs, p, fetch=False); eng.cluster(p, rank, source=0, fetch=False)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.collect(bs, p, fetch=False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(reps):
        eng.cluster(p, rank, source=0, fetch=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = eng.stats()
    print("%7d reads (%d records, %d signatures, %d clusters): collect %.3f ms/call, cluster %.3f ms/call" % (
        n, b.n_rec, st["n_sig"], st["n_clusters"], 1e3 * (t1 - t0) / reps, 1e3 * (t2 - t1) / reps))

# the per-read entry points of the drop-in, one call per record against one batched call (tests/golden: the fuzzA SAM text)
from tests import helpers as H                             # noqa: E402
import svim_amd                                           # noqa: E402
from svim_amd import records                              # noqa: E402
g2 = H.load("g2_collect.json.gz")
text = [c for c in g2["cases"] if c["name"] == "fuzzA" and c["mode"] == "coordinate" and c.get("sam")][0]["sam"]
bam = records.AlignmentFile(text=text)
recs = list(bam.fetch(until_eof=True))[:200]
oo = H.options({})
svim_amd.analyze_alignment_indel(recs[0], bam, recs[0].query_name, oo)
t0 = time.perf_counter()
one = [svim_amd.analyze_alignment_indel(a, bam, a.query_name, oo) for a in recs]
t1 = time.perf_counter()
many = svim_amd.analyze_alignment_indel_batch(recs, bam, [a.query_name for a in recs], oo)
t2 = time.perf_counter()
assert [[s.as_string() if hasattr(s, "as_string") else repr(s) for s in x[0]] for x in one] == [[s.as_string() if hasattr(s, "as_string") else repr(s) for s in x[0]] for x in many]
print("analyze_alignment_indel, %d records: %.2f ms per call one by one (%.1f ms in all), %.1f ms for ONE analyze_alignment_indel_batch call" % (
    len(recs), 1e3 * (t1 - t0) / len(recs), 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
