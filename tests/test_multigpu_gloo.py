"""world_size-4 gloo tests of the contig-sharded multi-GPU step (svim_amd/multigpu.py) on CPU tensors, the oracle standing in for the
GPU engine: contig ownership, the exchange of foreign signatures, the random.sample stream positions across ranks (the product finds them inside
svx_cluster - svx_cluster_set_ranks -, the oracle is told them through svo_cluster_set_chain by HostAdapter, which replays CPython's generator) and the final gather must reproduce the single-process result - cluster
records bit-identical, member lists identical as sets of emission keys in order."""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _setup(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _compare(res, full, full_keys):
    """rank 0: StepResult vs the single-process ClusterTable (members compared through the global emission keys)"""
    from svim_amd._abi import CLU_DTYPES
    got = res.to_host()
    if got.n != full.n or list(got.type_count) != list(full.type_count):
        return "n %d/%d type_count %r/%r" % (got.n, full.n, got.type_count, full.type_count)
    for k in CLU_DTYPES:
        a, b = getattr(got, k), getattr(full, k)[:full.n]
        same = (np.isnan(a) & np.isnan(b)) | (a == b) if a.dtype == np.float64 else a == b
        if not same.all():
            i = int(np.nonzero(~same)[0][0])
            return "%s[%d]: %r != %r" % (k, i, a[i], b[i])
    if not np.array_equal(got.member_off, full.member_off[:full.n + 1]):
        return "member_off differs"
    keys = res.sig_cols["key"].numpy()
    if full_keys is None:
        # emission keys are only comparable by ORDER (sharded file reading numbers its slots by region spans): position in emission
        # order of every gathered signature vs the single-process list index
        pos = np.empty(keys.size, dtype=np.int64)
        pos[np.argsort(keys, kind="stable")] = np.arange(keys.size)
        if not np.array_equal(pos[got.members], full.members[:full.n_members]):
            return "member positions differ"
    elif not np.array_equal(keys[got.members], full_keys[full.members[:full.n_members]]):
        return "member keys differ"
    return "ok"


def _worker_signatures(rank, world, port, ret, windows=False):
    """Signature-level: g5's 'stress31' list (3 contigs, 35 partitions beyond 100 members, all six types) dealt out to 4 ranks by
    contig owner; BND / DUP_INT rows are 'collected' by the owner of their OTHER contig, i.e. arrive as foreign rows.  With 3 contigs
    one rank owns nothing and only relays the stream positions."""
    _setup(rank, world, port)
    try:
        import helpers as H
        from oracle import oracle as om
        from svim_amd import _abi, batch, convert, multigpu
        g5 = H.load("g5_cluster.json.gz")
        case = [c for c in g5["cases"] if c["name"] == "stress31"][0]
        o = H.options(case["options"])
        sigs = [H.row_sig(r) for r in case["signatures"]]
        tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(g5["references"]))
        orc = om.Oracle()
        off, codes = convert.genome_arrays(o.genome, contigs.names)
        orc.set_genome(off, codes)
        p = _abi.Params.from_options(o)
        crank = batch.contig_ranks(contigs.names)
        full = orc.cluster(p, crank, table=tab)
        n = tab.n
        other = np.where(tab.type[:n] == _abi.SVX_DUP_INT, tab.contig[:n], np.where(tab.contig2[:n] >= 0, tab.contig2[:n], tab.contig[:n]))
        if windows:
            # coordinate windows (round 6): cuts proposed by length, i.e. INSIDE the contigs; a row is "collected" by the rank whose proposed window holds its START
            # (as a read near a cut would be) - once the cuts have moved into corridors some of those rows are foreign
            owner = multigpu.assign_windows(contigs.names, g5["lengths"], world)
            collector = owner.owner_of_positions(np.where(tab.contig2[:n] >= 0, other, tab.contig[:n]).astype(np.int64), tab.start[:n].astype(np.int64))
        else:
            owner = multigpu.assign_contigs(contigs.names, g5["lengths"], world)
            collector = owner[other]                                           # who "collected" the row
        idx = np.nonzero(collector == rank)[0]
        local = _abi.SigTable(len(idx), int((tab.seq_off[idx + 1] - tab.seq_off[idx]).sum()))
        for k in _abi.SIG_DTYPES:
            getattr(local, k)[:] = getattr(tab, k)[idx]
        ln = tab.seq_off[idx + 1] - tab.seq_off[idx]
        local.seq_off[1:] = np.cumsum(ln)
        pos = 0
        for i, l in zip(idx, ln):
            local.seq[pos:pos + l] = tab.seq[tab.seq_off[i]:tab.seq_off[i] + l]
            pos += int(l)
        if windows:
            n_foreign = -1                                                     # (known once the cuts are refined: rank 0 reports it from the result)
        else:
            n_foreign = int((owner[multigpu.owner_contig(local.type, local.contig, local.contig2)] != rank).sum()) if local.n else 0
        ad = multigpu.HostAdapter(orc, local)
        multigpu.wire_reset()
        res = multigpu.cluster_step(ad, p, rank, world, np.arange(len(contigs.names)), crank, owner)
        wire_full = dict(multigpu.WIRE)
        ret["chain%d" % rank] = max(ad.stream_end()) if ad.stream_end() else -1
        # the same step with the signature columns left on their ranks (bench.py --scaling strong: cluster rows + member lists only): identical clusters, one
        # gather less on the wire and fewer bytes; every collective of the step is in multigpu.WIRE
        ad2 = multigpu.HostAdapter(orc, local)
        multigpu.wire_reset()
        res2 = multigpu.cluster_step(ad2, p, rank, world, np.arange(len(contigs.names)), crank, owner, gather_signatures=False)
        wire_lean = dict(multigpu.WIRE)
        lean_ok = wire_lean["collectives"] == wire_full["collectives"] - 1 and 0 < wire_lean["bytes"] < wire_full["bytes"] and any("gather(final" in k for k in wire_lean["by_kind"])
        if rank == 0:
            verdict = _compare(res, full, tab.key[:n].astype(np.int64))
            if verdict == "ok":
                a, b = res.to_host(), res2.to_host()
                same = a.n == b.n and list(a.type_count) == list(b.type_count) and np.array_equal(a.members, b.members) and np.array_equal(a.member_off, b.member_off) and \
                    np.array_equal(a.start, b.start) and np.array_equal(a.score, b.score) and res2.sig_cols is None and list(res2.sig_counts) == list(res.sig_counts)
                verdict = "ok" if same else "gather_signatures=False changes the clusters"
            if verdict == "ok" and not lean_ok:
                verdict = "fabric accounting: %r vs %r" % (wire_lean, wire_full)
            ret[0] = verdict if verdict != "ok" else ("ok" if sum(res.sig_counts) == n and orc.stats()["n_large_partitions"] >= 0 else "counts")
        else:
            ret[rank] = "ok" if lean_ok else "fabric accounting: %r vs %r" % (wire_lean, wire_full)
        ret["foreign%d" % rank] = n_foreign
        if windows:
            if rank == 0:
                W = res.windows                                                 # the refined cuts every rank used
                own = W.owner_of_signatures(tab.type[:n], tab.contig[:n], tab.contig2[:n], tab.start[:n], tab.end[:n], tab.pos2[:n])
                ret["rows_per_rank"] = [int((own == r).sum()) for r in range(world)]
                ret["cuts"] = [(contigs.names[int(c)], int(x)) for c, x in zip(W.cut_contig, W.cut_pos)]
                ret["foreign_total"] = int((own != collector).sum())
                # no partition straddles a cut: single-process partitions of the whole list, owner of every member
                sidx, pid = orc.form_partitions(tab, crank, int(p.partition_max_distance))
                po = own[sidx]
                ret["straddling_partitions"] = int(sum(1 for a, b in zip(*_runs(pid)) if po[a:b].min() != po[a:b].max()))
        else:
            ret["owned%d" % rank] = int((owner == rank).sum())
    finally:
        dist.destroy_process_group()


def _worker_records(rank, world, port, ret, windows=False):
    """Record-level: g2's split-read fuzz set (3 contigs, BNDs and DUP_INTs across contigs, secondary / low-mapq records) collected
    per rank from the records of its contigs with GLOBAL emission slots; read ids are rank-local and foreign rows travel with their
    read names."""
    _setup(rank, world, port)
    try:
        import helpers as H
        from oracle import oracle as om
        from svim_amd import _abi, batch, convert, multigpu, records
        g = H.load("g2_collect.json.gz")
        case = [c for c in g["cases"] if c["name"] == "fuzzA" and c["mode"] == "coordinate" and c.get("sam")][0]
        o = H.options(case["options"])
        if windows:
            # coordinate windows: a sparser, longer input than the fuzz set (planted sites every ~8 kb + split reads: corridors are plentiful, as on a genome)
            from svim_amd import synth
            refs, lens = ["chr1", "chr2", "chr10"], [420000, 160000, 150000]
            ref = synth.make_reference(11, list(zip(refs, lens)))
            rr = synth.planted_reads(12, 900, ref, refs, lens, n_sites=90, types=("DEL", "INS", "INV")) + synth.fuzz_split_reads(13, 80, refs, lens)
            bam = records.AlignmentFile(text=synth.sam_text(refs, lens, synth.coordinate_sort(rr)))
            o.genome = ref
            o.max_sv_size = 3000           # (a 90 kb deletion between two fuzz segments spans a fifth of chr1: no cut may fall inside a signature, and the balance this test asserts would be its doing)
        else:
            bam = records.AlignmentFile(text=case["sam"])
        recs = list(bam.fetch(until_eof=True))
        refs, lens = list(bam.references), list(bam.lengths)
        p = _abi.Params.from_options(o)
        orc = om.Oracle()
        off, codes = convert.genome_arrays(o.genome, refs)
        orc.set_genome(off, codes)
        hb_all = batch.build_batch(bam, o, mode="coordinate")
        sig_all, _ = orc.collect(hb_all, p)
        full = orc.cluster(p, hb_all.contig_rank, table=sig_all)
        if windows:
            # proposed cuts balance the RECORDS (what a driver knows before COLLECT: the index of a BAM file, here a histogram of the record starts per 10 kb)
            bins = 10000
            dens = [np.bincount([a.reference_start // bins for a in recs if a.reference_id == k], minlength=-(-lens[k] // bins)).astype(np.float64) for k in range(len(refs))]
            owner = multigpu.assign_windows(refs, lens, world, weights=dens, bin_size=bins)
            rec_owner = owner.owner_of_positions(np.asarray([max(a.reference_id, 0) for a in recs], dtype=np.int64), np.asarray([a.reference_start for a in recs], dtype=np.int64))
            mine = [i for i, a in enumerate(recs) if (rec_owner[i] if a.reference_id >= 0 else world - 1) == rank]
        else:
            owner = multigpu.assign_contigs(refs, lens, world)
            # with 3 contigs and 4 ranks one rank stays empty; unplaced records (tid -1) go to the last rank like the file tail
            mine = [i for i, a in enumerate(recs) if (owner[a.reference_id] if a.reference_id >= 0 else world - 1) == rank]
        hb = batch.build_batch(bam, o, mode="coordinate", records=[recs[i] for i in mine])
        gi = np.asarray(mine, dtype=np.int64)
        hb.arrays["order"] = (2 * gi).astype(np.uint32)                    # emission slots in FILE order, not in local order
        hb.arrays["seg_order"] = (2 * gi + 1).astype(np.uint32)
        sig, _ = orc.collect(hb, p)
        names = list(hb.read_names)
        index = {nm: i for i, nm in enumerate(names)}

        def names_of(ids):
            return [names[int(i)] for i in ids]

        def ids_of(nms):
            out = []
            for nm in nms:
                if nm not in index:
                    index[nm] = len(names)
                    names.append(nm)
                out.append(index[nm])
            return out
        ad = multigpu.HostAdapter(orc, sig)
        res = multigpu.cluster_step(ad, p, rank, world, np.arange(len(refs)), hb_all.contig_rank, owner, names_of=names_of, ids_of=ids_of)
        if windows:
            n_foreign = -1
            if rank == 0:
                W, t, n = res.windows, sig_all, sig_all.n
                own = W.owner_of_signatures(t.type[:n], t.contig[:n], t.contig2[:n], t.start[:n], t.end[:n], t.pos2[:n])
                ret["rows_per_rank"] = [int((own == r).sum()) for r in range(world)]
                ret["cuts"] = [(refs[int(c)], int(x)) for c, x in zip(W.cut_contig, W.cut_pos)]
                sidx, pid = orc.form_partitions(t, hb_all.contig_rank, int(p.partition_max_distance))
                po = own[sidx]
                ret["straddling_partitions"] = int(sum(1 for a, b in zip(*_runs(pid)) if po[a:b].min() != po[a:b].max()))
        else:
            n_foreign = int((owner[multigpu.owner_contig(sig.type[:sig.n], sig.contig[:sig.n], sig.contig2[:sig.n])] != rank).sum()) if sig.n else 0
        ret["foreign%d" % rank] = n_foreign
        if rank == 0:
            verdict = _compare(res, full, sig_all.key[:sig_all.n].astype(np.int64))
            if verdict == "ok":
                # the gathered signature table, put back into emission order, is the single-process COLLECT result (read ids are
                # rank-local numbers: compared through nothing here, the clusters above depend on them)
                k = res.sig_cols["key"].numpy()
                order = np.argsort(k, kind="stable")
                for col in ("key", "type", "src", "aux", "contig", "start", "end", "contig2", "pos2"):
                    a = res.sig_cols[col].numpy()[order]
                    b = getattr(sig_all, col)[:sig_all.n]
                    if not np.array_equal(a, b.view(np.int64) if b.dtype == np.uint64 else b):
                        verdict = "signature column %s differs" % col
                        break
            ret[0] = verdict
        else:
            ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _runs(pid):
    """(starts, ends) of the runs of equal values in pid"""
    cut = np.nonzero(np.diff(pid))[0] + 1
    return np.concatenate([[0], cut]), np.concatenate([cut, [len(pid)]])


def _run(worker, world=4, extra=()):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, port, ret) + tuple(extra), nprocs=world, join=True)
    return dict(ret)


def test_eight_ranks_coordinate_windows_split_contigs():
    """Round 6: ranks own coordinate WINDOWS - cuts inside contigs, moved into corridors wider than partition_max_distance that no
    signature touches.  g5's stress31 list (3 contigs of 180 / 60 / 60 kb, 12.5 k signatures, 35 partitions beyond 100 members) on 8 ranks: every contig is
    split, the merged result equals the single-process oracle's (clusters, members, order, the random.sample streams relayed from rank to rank)."""
    ret = _run(_worker_signatures, world=8, extra=(True,))
    assert [ret[r] for r in range(8)] == ["ok"] * 8, ret
    cuts = ret["cuts"]
    assert sum(1 for c, x in cuts if x > 0) >= 5, cuts                      # cuts inside contigs ...
    assert len({c for c, x in cuts if x > 0}) == 3, cuts                    # ... of all three of them
    assert ret["straddling_partitions"] == 0
    rows = ret["rows_per_rank"]
    assert sum(1 for x in rows if x > 0) >= 2 and sum(rows) == 12525, rows  # (12.5 k rows in 300 kb: hardly a corridor - cuts that found none fell to an edge, the balance is what the list allows)
    assert ret["foreign_total"] > 0                                         # rows collected on one side of a moved cut and owned on the other
    ends = [ret["chain%d" % r] for r in range(8)]
    assert ends == sorted(ends) and ends[-1] > 1500


def test_eight_ranks_coordinate_windows_from_records_balanced():
    """The same on records: 980 reads on 3 contigs (420 / 160 / 150 kb) dealt out by the proposed windows of their start coordinate, every rank collects its
    own records (oracle), reads that span a cut leave rows on both sides (foreign rows with read names).  Merged == single process, and the rows every rank ends
    up clustering are balanced (the proposal balances the records per window from a histogram of their start coordinates)."""
    ret = _run(_worker_records, world=8, extra=(True,))
    assert [ret[r] for r in range(8)] == ["ok"] * 8, ret
    cuts, rows = ret["cuts"], ret["rows_per_rank"]
    assert sum(1 for c, x in cuts if x > 0) >= 6, cuts
    assert ret["straddling_partitions"] == 0
    assert min(rows) > 0 and max(rows) / (sum(rows) / 8.0) < 1.4, rows


def test_four_ranks_signature_lists_with_foreign_rows_and_stream_relay():
    ret = _run(_worker_signatures)
    assert [ret[r] for r in range(4)] == ["ok"] * 4, ret
    assert sum(ret["foreign%d" % r] for r in range(4)) > 50                # BND / DUP_INT rows really crossed ranks
    assert sorted(ret["owned%d" % r] for r in range(4)) == [0, 1, 1, 1]    # one rank only relays
    ends = [ret["chain%d" % r] for r in range(4)]
    assert ends == sorted(ends) and ends[-1] > 1500                        # every rank continued the streams of the one before


def test_four_ranks_records_with_read_names():
    ret = _run(_worker_records)
    assert [ret[r] for r in range(4)] == ["ok"] * 4, ret
    assert sum(ret["foreign%d" % r] for r in range(4)) > 5


def test_two_ranks_signature_lists():
    ret = _run(_worker_signatures, world=2)
    assert [ret[r] for r in range(2)] == ["ok"] * 2, ret


class _HostAccum(object):
    """The slice of svim_amd._lib.Engine that harness.BamPipeline drives (collect / accumulate / set_slot_base), on top of the oracle:
    the batches' tables are appended on the host with the slot base added to their keys."""

    def __init__(self, orc):
        from svim_amd import _abi
        self.orc, self._abi = orc, _abi
        self.parts, self.base, self.on = [], 0, False

    def accumulate(self, on):
        self.on = bool(on)
        if on:
            self.parts, self.base = [], 0

    def set_slot_base(self, base):
        self.base = int(base)

    def collect(self, b, p, fetch=False):
        sig, _ = self.orc.collect(b, p)
        sig.key = sig.key + (np.uint64(self.base) << np.uint64(32))
        self.parts.append(sig)

    def table(self):
        parts = self.parts
        n = sum(t.n for t in parts)
        out = self._abi.SigTable(n, sum(int(t.seq_off[t.n]) for t in parts))
        at, sq = 0, 0
        for t in parts:
            for k in self._abi.SIG_DTYPES:
                getattr(out, k)[at:at + t.n] = getattr(t, k)[:t.n]
            m = int(t.seq_off[t.n])
            out.seq_off[at:at + t.n + 1] = t.seq_off[:t.n + 1] + sq
            out.seq[sq:sq + m] = t.seq[:m]
            at += t.n
            sq += m
        return out


def _worker_bam(rank, world, port, ret, bam_path, fasta_path):
    """An indexed BAM with four contigs whose header order differs from the name order: every rank owns two NON-adjacent reference ids,
    reads them as two separate file regions through the .bai (svx_bam_seek), collects batch by batch, and the region sizes are
    exchanged so that the emission order is the file's."""
    _setup(rank, world, port)
    try:
        import helpers as H
        from oracle import oracle as om
        from svim_amd import _abi, batch, convert, harness, multigpu, records
        o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 20000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5,
                       "partition_max_distance": 1000, "position_distance_normalizer": 900, "edit_distance_normalizer": 1.0,
                       "cluster_max_distance": 0.5, "all_bnds": False})
        o.genome = fasta_path
        p = _abi.Params.from_options(o)
        bam = records.AlignmentFile(bam_path)
        refs = list(bam.references)
        orc = om.Oracle()
        off, codes = convert.genome_arrays(fasta_path, refs)
        orc.set_genome(off, codes)
        hb_all = batch.build_batch(bam, o, mode="coordinate")
        sig_all, _ = orc.collect(hb_all, p)
        full = orc.cluster(p, hb_all.contig_rank, table=sig_all)
        eng = _HostAccum(orc)

        class Adapter(multigpu.HostAdapter):
            def __init__(self):
                multigpu.HostAdapter.__init__(self, orc, None)

            def collect_counts(self):
                self.sig = eng.table()
                return multigpu.HostAdapter.collect_counts(self)
        res, pipe, names = harness.collect_cluster_bam_sharded(bam_path, o, eng, Adapter(), rank, world, threads=2, batch_records=60)
        ret["regions%d" % rank] = len(pipe.region_slots)
        ret["batches%d" % rank] = pipe.stats["batches"]
        pipe.bam.close()
        if rank == 0:
            verdict = _compare(res, full, None)
            if verdict == "ok":
                # every member signature of the gathered result resolves to its read NAME on rank 0 (ids are rank-local numbers shifted per rank)
                keys = res.sig_cols["key"].numpy()
                order = np.argsort(keys, kind="stable")
                got = [res.read_name(i) for i in res.sig_cols["read_id"].numpy()[order]]
                want = [hb_all.read_names[i] for i in sig_all.read_id[:sig_all.n]]
                if got != want:
                    verdict = "read names of the gathered signatures differ"
            ret[0] = verdict
        else:
            ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_ranks_indexed_bam_contig_runs(tmp_path):
    sys.path.insert(0, os.path.dirname(HERE))
    from svim_amd import records, synth
    refs, lens = ["chr1", "chr2", "chr10", "chr3"], [100000, 80000, 80000, 60000]
    ref = synth.make_reference(61, list(zip(refs, lens)))
    recs = synth.coordinate_sort(synth.fuzz_split_reads(62, 260, refs, lens, max_sv_size=20000) +
                                 synth.planted_reads(63, 300, ref, refs, lens, n_sites=25, types=("DEL", "INS", "INV")))
    bam_path, fa = str(tmp_path / "m.bam"), str(tmp_path / "m.fa")
    records.write_bam(bam_path, refs, lens, recs)
    synth.write_fasta(fa, ref)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_bam, args=(2, port, ret, bam_path, fa), nprocs=2, join=True)
    ret = dict(ret)
    assert [ret[r] for r in range(2)] == ["ok"] * 2, ret
    assert ret["regions0"] == 2 and ret["regions1"] == 2 and ret["batches0"] >= 3


def test_eight_ranks_five_of_them_empty():
    """The node the scaling bench runs on has eight GPUs: eight ranks over three contigs - five ranks own nothing, take part in every exchange and
    relay the stream positions untouched - still reproduce the single-process result on rank 0."""
    ret = _run(_worker_signatures, world=8)
    assert [ret[r] for r in range(8)] == ["ok"] * 8, ret
    assert sum(1 for r in range(8) if ret["owned%d" % r] > 0) == 3
    ends = [ret["chain%d" % r] for r in range(8)]
    assert ends == sorted(ends) and ends[-1] > 1500


def _worker_failing_rank(rank, world, port, ret, where):
    """One rank fails in the local work of a phase (the adapter's fetch raises): every rank must leave cluster_step with an exception -
    the failing one with its own, the others with RankFailed - instead of waiting in a collective the failed rank never enters."""
    _setup(rank, world, port)
    try:
        import helpers as H
        from oracle import oracle as om
        from svim_amd import _abi, batch, convert, multigpu
        g5 = H.load("g5_cluster.json.gz")
        case = [c for c in g5["cases"] if c["name"] == "stress31"][0]
        o = H.options(case["options"])
        sigs = [H.row_sig(r) for r in case["signatures"]]
        tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(g5["references"]))
        orc = om.Oracle()
        off, codes = convert.genome_arrays(o.genome, contigs.names)
        orc.set_genome(off, codes)
        p = _abi.Params.from_options(o)
        crank = batch.contig_ranks(contigs.names)
        owner = multigpu.assign_contigs(contigs.names, g5["lengths"], world)
        n = tab.n
        other = np.where(tab.type[:n] == _abi.SVX_DUP_INT, tab.contig[:n], np.where(tab.contig2[:n] >= 0, tab.contig2[:n], tab.contig[:n]))
        idx = np.nonzero(owner[other] == rank)[0]
        local = _abi.SigTable(len(idx), int((tab.seq_off[idx + 1] - tab.seq_off[idx]).sum()))
        for k in _abi.SIG_DTYPES:
            getattr(local, k)[:] = getattr(tab, k)[idx]
        ln = tab.seq_off[idx + 1] - tab.seq_off[idx]
        local.seq_off[1:] = np.cumsum(ln)
        pos = 0
        for i, l in zip(idx, ln):
            local.seq[pos:pos + l] = tab.seq[tab.seq_off[i]:tab.seq_off[i] + l]
            pos += int(l)

        class Failing(multigpu.HostAdapter):
            calls = 0

            def fetch_signatures(self, with_seq=True):
                Failing.calls += 1
                if rank == 1 and where == "fetch%d" % Failing.calls:
                    raise ValueError("planted failure on rank 1")
                return multigpu.HostAdapter.fetch_signatures(self, with_seq)

            def fetch_clusters(self):
                if rank == 1 and where == "clusters":
                    raise ValueError("planted failure on rank 1")
                return multigpu.HostAdapter.fetch_clusters(self)
        try:
            multigpu.cluster_step(Failing(orc, local), p, rank, world, np.arange(len(contigs.names)), crank, owner)
            ret[rank] = "returned"
        except ValueError as e:
            ret[rank] = "own: %s" % e
        except multigpu.RankFailed as e:
            ret[rank] = "peer"
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_takes_every_rank_out_of_the_step():
    import pytest
    for where in ("fetch1", "clusters"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_worker_failing_rank, args=(3, port, ret, where), nprocs=3, join=True)
        ret = dict(ret)
        assert ret[1].startswith("own:") and ret[0] == "peer" and ret[2] == "peer", (where, ret)


def test_eight_ranks_whole_genome_like_bam(tmp_path):
    """configs[3] on CPU: the 47-contig whole-genome profile (svim_amd/workloads.py c3: header order chr1..chr22, X, Y, M, scaffolds, HLA alleles - the name
    order interleaves them, so every rank reads SEVERAL non-adjacent regions of the file through the .bai) written by the device-side BAM writer with its
    index, eight ranks, the oracle as engine: merged result = single process, read names resolved on rank 0."""
    import torch
    sys.path.insert(0, os.path.dirname(HERE))
    from svim_amd import harness, synth, workloads
    prof = workloads.profile("c3", 0.001)
    prof["reads_per_mb"], prof["sites_per_mb"] = 260, 60
    b, genome, g_off, meta = workloads.make_batch_full(prof, seed=7, device="cpu")
    refs = [c[0] for c in prof["contigs"]]
    lens = [int(x) for x in (g_off[1:] - g_off[:-1]).tolist()]
    bam_path, fa = str(tmp_path / "wg.bam"), str(tmp_path / "wg.fa")
    n, _ = harness.write_bam_from_device_batch(bam_path, b, refs, lens, index=True, slab_bytes=8 << 20, threads=4)
    assert n == b.n_rec and os.path.exists(bam_path + ".bai")
    code = "=ACMGRSVTWYHKDBN"
    g = genome.numpy()
    synth.write_fasta(fa, {r: "".join(code[c] for c in g[int(g_off[i]):int(g_off[i + 1])].tolist()) for i, r in enumerate(refs)})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_bam, args=(8, port, ret, bam_path, fa), nprocs=8, join=True)
    ret = dict(ret)
    assert [ret[r] for r in range(8)] == ["ok"] * 8, ret
    assert sum(ret["regions%d" % r] for r in range(8)) >= 20          # the name order cuts the header order into many runs


def _worker_transport(rank, world, port, ret):
    """multigpu.TorchAllGather on raw host buffers (gather_into: what Engine.set_ranks hands libsvx's rank exchange to) and as bytes -> bytes"""
    import ctypes
    _setup(rank, world, port)
    from svim_amd import multigpu as MG
    try:
        ag = MG.TorchAllGather(world)
        ok = True
        for nbytes in (8, 128, 4099, 70001):
            mine = bytes((rank * 37 + i * 11) & 255 for i in range(nbytes))
            want = b"".join(bytes((r * 37 + i * 11) & 255 for i in range(nbytes)) for r in range(world))
            send = (ctypes.c_uint8 * nbytes).from_buffer_copy(mine)
            recv = (ctypes.c_uint8 * (nbytes * world))()
            ag.gather_into(ctypes.addressof(send), ctypes.addressof(recv), nbytes)
            ok = ok and bytes(recv) == want and ag(mine) == want
        ret[rank] = "ok" if ok and ag.calls == 8 else "transport returned other bytes"
    finally:
        dist.destroy_process_group()


def test_all_gather_transport_on_raw_buffers_and_bytes():
    out = _run(_worker_transport, world=3)
    assert out == {0: "ok", 1: "ok", 2: "ok"}
