import os, sys, types, itertools
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from svim_amd import _abi, _lib, devsynth
o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                          partition_max_distance=1000, position_distance_normalizer=900, edit_distance_normalizer=1.0, cluster_max_distance=0.5, all_bnds=False)
p = _abi.Params.from_options(o)
b, genome, meta = devsynth.make_batch(n_reads=1000000, contig_len=250_000_000, seed=2, device="cuda:0")
bs = b.struct()
for mp, blocks in itertools.product((0, 1), (4, 6, 8)):
    os.environ["SVX_SCAN_MAP"] = str(mp); os.environ["SVX_SCAN_BLOCKS"] = str(blocks)
    eng = _lib.Engine(0)
    ts = []
    for _ in range(4):
        eng.collect(bs, p, fetch=False)
        ts.append(eng.stats()["t_cigar_scan_ms"])
    print("map", mp, "blocks/CU", blocks, "scan ms", ["%.3f" % t for t in ts], "n_sig", eng.stats()["n_sig"])
    eng.close()
