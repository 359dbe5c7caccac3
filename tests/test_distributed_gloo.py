"""world_size-2 gloo test of the multi-GPU path on CPU: record sharding -> all-gather of signature tables ->
partition-sharded clustering -> gather + merge.  The oracle stands in for the GPU engine (tests may use it)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import helpers as H
        from oracle import oracle as om
        from svim_amd import _abi, batch, convert, records
        from svim_amd.distributed import all_gather_signatures, gather_clusters
        g = H.load("g2_collect.json.gz")
        case = [c for c in g["cases"] if c["name"] == "planted" and c.get("sam")][0]
        o = H.options(case["options"])
        bam = records.AlignmentFile(text=case["sam"])
        recs = list(bam.fetch(until_eof=True))
        p = _abi.Params.from_options(o)
        orc = om.Oracle()
        refs = list(bam.references)
        off, codes = convert.genome_arrays(o.genome, refs)
        orc.set_genome(off, codes)
        # single-process answer
        hb_all = batch.build_batch(bam, o, mode="coordinate")
        sig_all, _ = orc.collect(hb_all, p)
        full = orc.cluster(p, hb_all.contig_rank, table=sig_all, shard=(0, 1))
        # sharded: contiguous record ranges
        lo, hi = rank * len(recs) // world, (rank + 1) * len(recs) // world
        hb = batch.build_batch(bam, o, mode="coordinate", records=recs[lo:hi])
        sig, _ = orc.collect(hb, p)
        # read ids are rank-local: re-intern through names before the exchange
        names = hb.read_names
        glob = {n: i for i, n in enumerate(hb_all.read_names)}
        sig.read_id = np.array([glob[names[i]] for i in sig.read_id], dtype=np.int32)
        everything = all_gather_signatures(sig)
        assert everything.first_difference(sig_all) in (None,) or everything.equal(sig_all, with_key=False)
        ct = orc.cluster(p, hb_all.contig_rank, table=everything, shard=(rank, world))
        merged = gather_clusters(ct, hb_all.contig_rank)
        ret[rank] = merged.first_difference(full) or "ok"
    finally:
        dist.destroy_process_group()


def test_two_rank_pipeline_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def _worker_by_origin(rank, world, port, ret):
    """Contig-sharded input (the multi-GPU bench layout): rank r collects the records of its contigs; only the fixed-width columns
    are exchanged, every rank keeps ITS inserted sequences (remote signatures carry empty sequence ranges) and owns the partitions
    whose first sorted member it produced."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import helpers as H
        from oracle import oracle as om
        from svim_amd import _abi, batch, convert, records
        from svim_amd.distributed import all_gather_bytes, concat_sig_tables, gather_clusters, _pack_sig, _unpack_sig
        g = H.load("g2_collect.json.gz")
        case = [c for c in g["cases"] if c["name"] == "planted" and c.get("sam")][0]
        o = H.options(case["options"])
        bam = records.AlignmentFile(text=case["sam"])
        recs = list(bam.fetch(until_eof=True))
        p = _abi.Params.from_options(o)
        orc = om.Oracle()
        refs = list(bam.references)
        off, codes = convert.genome_arrays(o.genome, refs)
        orc.set_genome(off, codes)
        hb_all = batch.build_batch(bam, o, mode="coordinate")
        glob = {n: i for i, n in enumerate(hb_all.read_names)}
        # rank 0: everything left of a cut, rank 1: everything right of it; the cut sits in the widest signature-free stretch near the
        # middle, reads with records on both sides are left out
        sig_all, _ = orc.collect(hb_all, p)
        pos = np.sort(np.concatenate([sig_all.start[:sig_all.n], sig_all.end[:sig_all.n]]).astype(np.int64))
        mid = pos[(pos > pos[len(pos) // 3]) & (pos < pos[2 * len(pos) // 3])]
        gaps = np.diff(mid)
        cut = int(mid[np.argmax(gaps)] + gaps.max() // 2)
        side = {}
        for a in recs:
            end = a.reference_end if a.reference_end is not None else a.reference_start + 1
            side.setdefault(a.query_name, set()).update({0 if a.reference_start < cut else 1, 0 if end <= cut else 1})
        mine = [a for a in recs if side[a.query_name] == {rank}]
        kept = [a for a in recs if len(side[a.query_name]) == 1]
        def collect(rs):
            hb = batch.build_batch(bam, o, mode="coordinate", records=rs)
            sig, _ = orc.collect(hb, p)
            sig.read_id = np.array([glob[hb.read_names[i]] for i in sig.read_id], dtype=np.int32)
            return sig
        sig = collect(mine)
        parts = [_unpack_sig(b) for b in all_gather_bytes(_pack_sig(sig))]
        # reference answer: everything on one process, rank-major order
        full_tab = concat_sig_tables(parts)
        full = orc.cluster(p, hb_all.contig_rank, table=full_tab, shard=(0, 1))
        # fast path table: remote sequences are NOT available
        for r, t in enumerate(parts):
            if r != rank:
                t.seq_off = np.zeros(t.n + 1, dtype=np.int64)
                t.seq = np.zeros(1, dtype=np.uint8)
        local = concat_sig_tables(parts)
        prefix = np.concatenate([[0], np.cumsum([t.n for t in parts])]).astype(np.int64)
        ct = orc.cluster(p, hb_all.contig_rank, table=local, shard=(rank, world), origin_prefix=prefix)
        remote = orc.remote_members()
        merged = gather_clusters(ct, hb_all.contig_rank)
        # the device-side exchange of bench.py (all-gather of the columns + torch merge), here on CPU tensors over gloo
        import torch
        from svim_amd._abi import CLU_DTYPES
        from svim_amd.distributed import _all_gather_var, merge_gathered_clusters
        cnt = torch.tensor([ct.n, ct.n_members], dtype=torch.int64)
        allc = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allc, cnt)
        ns, nms = [int(c[0]) for c in allc], [int(c[1]) for c in allc]
        g = {k: _all_gather_var(torch.from_numpy(np.ascontiguousarray(getattr(ct, k)[:ct.n])), ns, dist, torch) for k in CLU_DTYPES}
        g_part = _all_gather_var(torch.from_numpy(np.ascontiguousarray(ct.part_index[:ct.n]).astype(np.int64)), ns, dist, torch)
        g_mem = _all_gather_var(torch.from_numpy(np.ascontiguousarray(ct.members[:ct.n_members])), nms, dist, torch)
        merged_t = merge_gathered_clusters(g, g_part, g_mem, hb_all.contig_rank).to_host()
        same_t = merged_t.first_difference(full) is None and np.array_equal(merged_t.part_index, merged.part_index)
        ok = merged.first_difference(full) is None and same_t and remote == 0 and merged.n > 20 and len(kept) > len(mine) > 0
        ret[rank] = "ok" if ok else "diff %s remote %d n %d" % (merged.first_difference(full), remote, merged.n)
    finally:
        dist.destroy_process_group()


def test_two_rank_by_origin_fast_path_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_by_origin, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
