"""GPU parity tests (run on a real MI355X with `-m gpu`): the HIP path, called through the C ABI
(svim_amd/libsvx.so), against (a) the golden vectors produced by running the reference and (b) the oracle
on the same seeded inputs.  Bit-exact for every integer / index / byte result; FP64 scores within 1e-9
relative of the reference (north_star tolerance: 1e-6) and 1e-12 of the oracle."""
import random
import struct

import numpy as np
import pytest

import helpers as H
from svim_amd import _abi, batch, convert, records, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from svim_amd import _lib
    return _lib.Engine(0)


def test_cigar_indel_golden(eng):
    g = H.load("g1_cigar_indel.json.gz")
    for c in g["cases"]:
        got = eng.cigar_indel([tuple(t) for t in c["tuples"]], c["min_length"])
        assert got == [tuple(x) for x in c["expect"]], c["tuples"]


def test_edit_distance_golden(eng):
    g = H.load("g_editdistance.json.gz")
    pairs = [(a, b) for a, b, d in g["cases"]] + [(b, a) for a, b, d in g["cases"]]
    exp = [d for a, b, d in g["cases"]] * 2
    assert eng.edit_distances(pairs) == exp


@pytest.mark.parametrize("few_pairs", ["0", "2048"])
@pytest.mark.parametrize("alphabet", ["ACGTN", "ACGT"])      # generic 4-plane kernels / 2-plane A,C,G,T kernels
def test_edit_distance_long_vs_oracle(eng, oracle, alphabet, few_pairs, monkeypatch):
    monkeypatch.setenv("SVX_EDIT_FEW_PAIRS", few_pairs)
    rng = random.Random(3)
    pairs = []
    for la, lb, sim in ((2047, 2049, True), (2100, 2300, True), (4000, 4100, False), (5000, 300, False),
                        (8200, 8300, True), (9000, 9100, False), (17000, 16500, True), (1, 5000, False), (0, 70, False),
                        (3000, 3000, True), (6000, 6500, False), (4100, 8000, False)):
        a = synth.random_seq(rng, la)
        if sim:
            b = list(a)
            for _ in range(max(1, la // 15)):
                p = rng.randrange(len(b))
                r = rng.random()
                if r < 0.4:
                    b[p] = rng.choice(alphabet)
                elif r < 0.7:
                    del b[p]
                else:
                    b.insert(p, rng.choice("ACGT"))
            b = "".join(b)
            b = (b + synth.random_seq(rng, lb))[:lb] if len(b) < lb else b[:lb]
        else:
            b = synth.random_seq(rng, lb)
        pairs.append((a, b))
    got = eng.edit_distances(pairs)
    exp = [oracle.edit_distance(a, b) for a, b in pairs]
    assert got == exp


@pytest.mark.parametrize("few_pairs", ["0", "2048"])            # band route as in a large call / low-latency route of a call with few pairs
@pytest.mark.parametrize("alphabet", ["ACGTN", "ACGT"])
def test_edit_distance_banded_classes_vs_oracle(eng, oracle, alphabet, few_pairs, monkeypatch):
    """Similar pairs with substitutions AND indels at several divergences / lengths / length differences: drives every
    band class (32..512 diagonals), the failed-band retry and the full-matrix fallback; also the '=' symbol (code 0)."""
    monkeypatch.setenv("SVX_EDIT_FEW_PAIRS", few_pairs)
    rng = random.Random(11)
    pairs = []
    for it in range(700):
        la = rng.choice((10, 40, 90, 200, 400, 800, 1500, 3000))
        a = synth.random_seq(rng, la)
        b = list(a)
        div = rng.choice((0.0, 0.01, 0.03, 0.08, 0.2, 0.5))
        for _ in range(int(div * la) + rng.choice((0, 0, 1, 5))):
            if not b:
                break
            p = rng.randrange(len(b))
            r = rng.random()
            if r < 0.4:
                b[p] = rng.choice(alphabet)
            elif r < 0.7:
                del b[p]
            else:
                b.insert(p, rng.choice("ACGT"))
        b = "".join(b)
        r = rng.random()
        if r < 0.15:
            b = b + synth.random_seq(rng, rng.choice((5, 40, 300)))      # length difference
        elif r < 0.25:
            b = synth.random_seq(rng, rng.choice((7, 33, 250))) + b
        if it % 97 == 0 and alphabet != "ACGT":
            a = a[:len(a) // 2] + "=" + a[len(a) // 2:]
        if rng.random() < 0.5:
            a, b = b, a
        pairs.append((a, b))
    got = eng.edit_distances(pairs)
    exp = [oracle.edit_distance(a, b) for a, b in pairs]
    bad = [(i, g, e, len(pairs[i][0]), len(pairs[i][1])) for i, (g, e) in enumerate(zip(got, exp)) if g != e]
    assert not bad, bad[:10]


def test_linkage_golden(eng):
    g = H.load("g_linkage.json.gz")
    by_t = {}
    for c in g["cases"]:
        by_t.setdefault(c["t"], []).append(c)
    for t, cases in by_t.items():
        probs = [(c["n"], np.array([float.fromhex(x) for x in c["d"]])) for c in cases]
        labs = eng.linkage_fcluster(probs, t)
        for c, lab in zip(cases, labs):
            assert lab.tolist() == c["labels"], (c["n"], t)


@pytest.mark.parametrize("idx", range(len(H.load("g2_collect.json.gz")["cases"])))
def test_collect_golden_and_oracle(eng, oracle, idx):
    g = H.load("g2_collect.json.gz")
    case = g["cases"][idx]
    bam, hb, o = H.sam_case_batch(case, g)
    p = _abi.Params.from_options(o)
    sig, bnd = eng.collect(hb, p)
    assert H.table_rows(sig, hb.references, hb.read_names) == case["signatures"]
    assert H.table_rows(bnd, hb.references, hb.read_names) == case["bnds"]
    osig, obnd = oracle.collect(hb, p)
    assert sig.first_difference(osig) is None
    assert bnd.first_difference(obnd) is None


@pytest.mark.parametrize("idx", range(len(H.load("g5_cluster.json.gz")["cases"])))
def test_cluster_golden_and_oracle(eng, oracle, idx):
    g = H.load("g5_cluster.json.gz")
    case = g["cases"][idx]
    o = H.options(case["options"])
    sigs = [H.row_sig(r) for r in case["signatures"]]
    tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(H.REFS))
    off, codes = convert.genome_arrays(o.genome, contigs.names)
    rank = batch.contig_ranks(contigs.names)
    p = _abi.Params.from_options(o)
    eng.set_genome(off, codes)
    ct = eng.cluster(p, rank, table=tab)
    H.compare_cluster_rows(H.cluster_rows(ct, contigs.names), case["clusters"])
    oracle.set_genome(off, codes)
    oc = oracle.cluster(p, rank, table=tab)
    assert ct.first_difference(oc, rtol=1e-12) is None


def test_form_partitions_on_the_gpu_match_reference_golden(eng):
    """form_partitions (src/svim/SVIM_clustering.py:17-29) as the GPU makes it - key sort + boundary flags + scan - against g4_partitions, the reference's own
    partitions of the g5 cases (so far only the oracle and a host helper saw this fixture; the GPU's partitions were pinned through the
    member lists of g5 only)."""
    g4 = H.load("g4_partitions.json.gz")
    g5 = H.load("g5_cluster.json.gz")
    cases = {c["name"]: c for c in g5["cases"]}
    done = {}
    checked = 0
    for p in g4["partitions"]:
        case = cases[p["case"]]
        if p["case"] not in done:
            o = H.options(case["options"])
            sigs = [H.row_sig(r) for r in case["signatures"]]
            tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(H.REFS))
            off, codes = convert.genome_arrays(o.genome, contigs.names)
            eng.set_genome(off, codes)
            eng.cluster(_abi.Params.from_options(o), batch.contig_ranks(contigs.names), table=tab, fetch=False)
            done[p["case"]] = (tab, eng.partitions())
        tab, parts = done[p["case"]]
        code = _abi.TYPE_CODE[p["type"]]
        got = [q for q in parts if tab.type[q[0]] == code]
        assert all(len(set(int(tab.type[i]) for i in q)) == 1 for q in parts)
        assert got == p["partitions"], (p["case"], p["type"])
        checked += len(got)
    assert checked > 50
    assert sorted(i for q in done[p["case"]][1] for i in q) == list(range(len(done[p["case"]][0].type)))


def test_sampling_many_partition_sizes_vs_oracle(eng, oracle):
    """random.sample replay: partitions of every size 101..190, around 256/512/1024 and the pool/set switch (1045/1046),
    several per type so that the RNG stream is carried across them."""
    rng = random.Random(5)
    sizes = list(range(101, 191)) + [250, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1024, 1025, 1044, 1045, 1046, 1047, 2000]
    rng.shuffle(sizes)
    rows = []
    pos = 10000
    rid = 0
    for n in sizes:
        typ = rng.choice(("DEL", "DEL", "INV", "DUP_TAN"))
        span = rng.choice((80, 300))
        for k in range(n):
            rid += 1
            st = pos + rng.randint(-150, 150)
            sp = span + rng.randint(-20, 20)
            if typ == "DEL":
                rows.append(["DEL", "chr1", st, st + sp, "cigar", "r%d" % rid])
            elif typ == "INV":
                rows.append(["INV", "chr1", st, st + sp, "suppl", "r%d" % rid, rng.choice(("left_fwd", "right_rev"))])
            else:
                rows.append(["DUP_TAN", "chr1", st, st + sp, "suppl", "r%d" % rid, 2, True])
        pos += 5000
    rng.shuffle(rows)
    sigs = [H.row_sig(r) for r in rows]
    tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(H.REFS))
    rank = batch.contig_ranks(contigs.names)
    p = _abi.Params.from_options(H.options({"partition_max_distance": 1000, "cluster_max_distance": 0.5, "position_distance_normalizer": 900,
                                            "edit_distance_normalizer": 1.0}))
    ct = eng.cluster(p, rank, table=tab)
    oc = oracle.cluster(p, rank, table=tab)
    assert ct.n == oc.n and ct.first_difference(oc, rtol=1e-12) is None


def _planted_case(seed, n_reads, n_sites):
    contigs = [("chr1", 180000), ("chr2", 60000), ("chr10", 60000)]
    refs = synth.make_reference(1, contigs)
    references = [c[0] for c in contigs]
    lengths = [c[1] for c in contigs]
    recs = synth.planted_reads(seed, n_reads, refs, references, lengths, n_sites=n_sites, types=("DEL", "INS", "INV"))
    recs += synth.fuzz_split_reads(seed + 1, n_reads // 10, references, lengths)
    text = synth.sam_text(references, lengths, synth.coordinate_sort(recs))
    return records.AlignmentFile(text=text), refs, references


def test_collect_then_cluster_resident_vs_oracle(eng, oracle):
    """COLLECT output stays in HBM and is clustered from there (source 0) - the bench path - and must equal the
    oracle run on the same batch."""
    bam, refs, references = _planted_case(77, 2500, 60)
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10,
                   "segment_overlap_tolerance": 5, "partition_max_distance": 1000, "position_distance_normalizer": 900,
                   "edit_distance_normalizer": 1.0, "cluster_max_distance": 0.5, "all_bnds": True})
    hb = batch.build_batch(bam, o, mode="coordinate")
    p = _abi.Params.from_options(o)
    off, codes = convert.genome_arrays(refs, references)
    eng.set_genome(off, codes)
    oracle.set_genome(off, codes)
    sig, bnd = eng.collect(hb, p)
    osig, obnd = oracle.collect(hb, p)
    assert sig.n > 1000
    assert sig.first_difference(osig) is None and bnd.first_difference(obnd) is None
    # the same batch with its arrays in page-locked memory of the library (svx_host_alloc: uploaded in place, no bounce pass): same tables;
    # the blocks go back to the library with the arrays and are handed out again
    hp = hb.pinned()
    psig, pbnd = eng.collect(hp, p)
    assert psig.first_difference(osig) is None and pbnd.first_difference(obnd) is None
    first = hp.arrays["cigar"].ctypes.data
    del hp, psig, pbnd
    import gc
    gc.collect()
    hp = hb.pinned()
    assert hp.arrays["cigar"].ctypes.data == first or any(a.ctypes.data == first for a in hp.arrays.values())
    sig, bnd = eng.collect(hp, p)
    assert sig.first_difference(osig) is None and bnd.first_difference(obnd) is None
    for source in (0, 1):
        ct = eng.cluster(p, hb.contig_rank, source=source)
        oc = oracle.cluster(p, hb.contig_rank, source=source)
        assert ct.n > 0
        assert ct.first_difference(oc, rtol=1e-12) is None



def test_dropin_api_matches_reference_golden(eng):
    """The reference-named entry points return objects equal to what the reference returned."""
    import svim_amd
    g = H.load("g2_collect.json.gz")
    case = [c for c in g["cases"] if c["name"] == "fuzzA" and c["mode"] == "coordinate" and not c["options"]["all_bnds"]][0]
    o = H.options(case["options"])
    bam = records.AlignmentFile(text=case["sam"])
    sigs, bnds = svim_amd.analyze_alignment_file_coordsorted(bam, o)
    assert [H.sig_row(s) for s in sigs] == case["signatures"]
    g5 = H.load("g5_cluster.json.gz")
    ccase = [c for c in g5["cases"] if c["name"] == "from_collect:fuzzA:coordinate"][0]
    res = svim_amd.cluster_sv_signatures(sigs, H.options(ccase["options"]))
    idx = {id(s): i for i, s in enumerate(sigs)}
    got = []
    for k, lst in enumerate(res):
        rows = []
        for c in lst:
            mem = [idx[id(m)] for m in c.members]
            if k < 3:
                rows.append([c.contig, c.start, c.end, c.score, c.size, c.std_span, c.std_pos, mem])
            else:
                row = [c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end, c.score, c.size,
                       c.std_span, c.std_pos, mem]
                if c.type == "BND":
                    row += [c.direction1, c.direction2]
                rows.append(row)
        got.append(rows)
    H.compare_cluster_rows(got, ccase["clusters"])
    # the reference's own known-answer vectors (src/tests/test_intra.py:9-22) through the drop-in name
    assert svim_amd.analyze_cigar_indel([(5, 10), (4, 20), (0, 30), (2, 40), (1, 50), (0, 30), (4, 25), (5, 15)], 30) == \
        [(30, 50, 40, "DEL"), (70, 50, 50, "INS")]


def test_multigpu_step_single_rank_device_path(eng, oracle):
    """bench.py's multi-GPU step (svim_amd/multigpu.py: SvxAdapter, everything device-resident, RCCL process group of one rank) must
    reproduce the plain single-GPU result, both when it clusters the resident table (no foreign rows: the bench layout) and when it
    is handed a table of device tensors (the route taken after an exchange of foreign rows); svx_cluster_stream_positions reports where the
    random.sample streams ended."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from svim_amd import multigpu as MG, workloads
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = "cuda:0"
        prof = workloads.profile("c2", 0.012)
        b, genome, g_off, meta = workloads.make_batch_full(prof, seed=5, device=dev)
        o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5,
                       "partition_max_distance": 5000, "position_distance_normalizer": 900, "edit_distance_normalizer": 1.0,
                       "cluster_max_distance": 0.5, "all_bnds": False})
        p = _abi.Params.from_options(o)
        crank = b.t["contig_rank"].cpu().numpy().astype(np.int32)
        eng.set_genome(g_off, genome, on_device=True)
        eng.collect(b.struct(), p, fetch=False)
        direct = eng.cluster(p, crank, source=0)
        sig = eng.fetch_signatures(0)
        ad = MG.SvxAdapter(eng, dev)
        gid = np.arange(len(crank))
        owner = np.zeros(len(crank), dtype=np.int32)
        eng.collect(b.struct(), p, fetch=False)
        res = MG.cluster_step(ad, p, 0, 1, gid, crank, owner)
        merged = res.to_host()
        assert merged.n > 100 and all(c > 0 for c in merged.type_count)
        assert merged.first_difference(direct) is None
        assert np.array_equal(res.sig_cols["key"].cpu().numpy().view(np.uint64), sig.key[:sig.n])
        # the table route: the same signatures handed over as device tensors
        cols, seq_off, seq = ad.fetch_signatures()
        ad.cluster(p, crank, table=(cols, seq_off, seq))
        again = eng.fetch_clusters()
        assert again.first_difference(direct) is None
        starts0, ends0 = eng.stream_positions()
        assert starts0 == [0] * 6 and all(e >= 0 for e in ends0)          # a single rank: every seed(1524) stream starts at its beginning
        # and the device-generated batch agrees with the oracle, stream positions included
        oracle.set_genome(g_off.cpu().numpy().astype(np.int64), genome.cpu().numpy())
        hb = b.slice_records(0, b.n_rec)
        osig, _ = oracle.collect(hb, p)
        olog = []
        oracle.set_chain(lambda phase, w: olog.append(list(w)))
        oc = oracle.cluster(p, hb.contig_rank, source=0)
        oracle.set_chain(None)
        assert sig.first_difference(osig) is None
        assert direct.first_difference(oc, rtol=1e-12) is None
        assert olog[1] == ends0                                          # both engines consumed the streams to the same positions
    finally:
        dist.destroy_process_group()


def test_bam_path_native_reader_route(eng, tmp_path):
    """analyze_alignment_file_*(path_to_bam, options): C++ BAM front-end -> batches -> svx_collect must return the same
    objects as the Python-reader route on the same records (both sort modes, several batches)."""
    import svim_amd
    from svim_amd import SVIM_COLLECT
    g = H.load("g2_collect.json.gz")
    for mode, fn in (("coordinate", svim_amd.analyze_alignment_file_coordsorted), ("queryname", svim_amd.analyze_alignment_file_querysorted)):
        case = [c for c in g["cases"] if c["name"] == "fuzzB" and c["mode"] == mode and c["options"]["all_bnds"]][0]
        text = [c for c in g["cases"] if c["name"] == "fuzzB" and c["mode"] == mode and c.get("sam")][0]["sam"]
        bam = records.AlignmentFile(text=text)
        recs = list(bam.fetch(until_eof=True))
        path = str(tmp_path / ("%s.bam" % mode))
        records.write_bam(path, bam.references, bam.lengths, recs, sort_order=mode)
        o = H.options(case["options"])
        sigs, bnds = fn(path, o)
        assert [H.sig_row(s) for s in sigs] == case["signatures"]
        assert [H.sig_row(s) for s in bnds] == case["bnds"]
        sigs2, bnds2 = SVIM_COLLECT._run_native(path, o, mode, batch_records=41)
        assert [H.sig_row(s) for s in sigs2] == case["signatures"]
        assert [H.sig_row(s) for s in bnds2] == case["bnds"]


def test_c1_config0_through_bam_file(eng, tmp_path):
    """BASELINE.json configs[0] end to end through the drop-in entry points: BAM file on disk -> native reader ->
    analyze_alignment_file_coordsorted -> cluster_sv_signatures (genome from a FASTA file) == the reference's output."""
    import svim_amd
    g, refs, recs = H.c1_case()
    path = str(tmp_path / "c1.bam")
    records.write_bam(path, ["chr1"], [2000000], recs)
    fa = str(tmp_path / "c1.fa")
    synth.write_fasta(fa, refs)
    o = H.options(g["options"])
    o.genome = fa
    sigs, bnds = svim_amd.analyze_alignment_file_coordsorted(path, o)
    assert [H.sig_row(s) for s in sigs] == g["signatures"]
    res = svim_amd.cluster_sv_signatures(sigs, o)
    idx = {id(s): i for i, s in enumerate(sigs)}
    got = []
    for k, lst in enumerate(res):
        rows = []
        for c in lst:
            mem = [idx[id(m)] for m in c.members]
            rows.append([c.contig, c.start, c.end, c.score, c.size, c.std_span, c.std_pos, mem] if k < 3 else
                        [c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end, c.score, c.size, c.std_span,
                         c.std_pos, mem])
        got.append(rows)
    H.compare_cluster_rows(got, g["clusters"])


def test_dropin_collect_to_cluster_stays_resident_and_lazy(eng, tmp_path):
    """COLLECT -> CLUSTER through the drop-in names without building a single Python object: the SignatureList returned by
    analyze_alignment_file_coordsorted still mirrors the table resident in HBM, so cluster_sv_signatures runs with source = 0
    (nothing is uploaded - checked through the call arguments) and hands back lazy cluster lists.  The Python time of the two
    entry points at configs[0] size is asserted to be well below what object conversion alone used to cost."""
    import time
    import svim_amd
    from svim_amd import _lib, lazy
    g, refs, recs = H.c1_case()
    path = str(tmp_path / "c1.bam")
    records.write_bam(path, ["chr1"], [2000000], recs)
    fa = str(tmp_path / "c1.fa")
    synth.write_fasta(fa, refs)
    o = H.options(g["options"])
    o.genome = fa
    svim_amd.cluster_sv_signatures(svim_amd.analyze_alignment_file_coordsorted(path, o)[0], o)      # warm: genome upload, allocations
    seen = []
    engine = _lib.engine()
    orig = engine.cluster

    def spy(*a, **kw):
        seen.append((kw.get("source", 2), kw.get("table")))
        return orig(*a, **kw)
    engine.cluster = spy
    try:
        t0 = time.perf_counter()
        sigs, bnds = svim_amd.analyze_alignment_file_coordsorted(path, o)
        t1 = time.perf_counter()
        res = svim_amd.cluster_sv_signatures(sigs, o)
        t2 = time.perf_counter()
    finally:
        engine.cluster = orig
    assert isinstance(sigs, lazy.SignatureList) and sigs._objs is None and not sigs._one
    assert seen == [(0, None)]                                           # clustered from the resident table
    assert all(isinstance(r, lazy.ClusterList) and r._objs is None for r in res)
    assert len(sigs) == len(g["signatures"]) and [len(r) for r in res] == [len(x) for x in g["clusters"]]
    assert t2 - t1 < 0.25, "cluster_sv_signatures on a resident table took %.3f s" % (t2 - t1)
    # the side list clusters from the device too
    seen.clear()
    engine.cluster = spy
    try:
        svim_amd.cluster_sv_signatures(bnds, o)
    finally:
        engine.cluster = orig
    assert seen == [(1, None)]
    # reading the results builds them, and they are the reference's
    assert [H.sig_row(s) for s in sigs] == g["signatures"]
    idx = {id(s): i for i, s in enumerate(sigs)}
    got = []
    for k, lst in enumerate(res):
        got.append([[c.contig, c.start, c.end, c.score, c.size, c.std_span, c.std_pos, [idx[id(m)] for m in c.members]] if k < 3 else
                    [c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end, c.score, c.size, c.std_span,
                     c.std_pos, [idx[id(m)] for m in c.members]] for c in lst])
    H.compare_cluster_rows(got, g["clusters"])
    # a list that no longer mirrors the device table (another COLLECT ran since) is uploaded as a table, still without objects
    sigs2, _ = svim_amd.analyze_alignment_file_coordsorted(path, o)
    svim_amd.analyze_alignment_file_coordsorted(path, o)
    seen.clear()
    engine.cluster = spy
    try:
        res2 = svim_amd.cluster_sv_signatures(sigs2, o)
    finally:
        engine.cluster = orig
    assert seen[0][0] == 2 and seen[0][1] is sigs2.table and sigs2._objs is None
    assert [len(r) for r in res2] == [len(x) for x in g["clusters"]]


def test_bam_pipeline_accumulates_batches_on_device(oracle, tmp_path):
    """svim_amd/harness.py:BamPipeline - reader thread ahead of the GPU thread, sparse SEQ, the signature lists of all batches
    appended in HBM (svx_collect_accumulate / svx_collect_set_slot_base) - must give what ONE batch holding the whole file gives:
    same signature rows in the same order, same clusters; and that equals the oracle on the file's records."""
    from svim_amd import _lib, harness, workloads
    from svim_amd.bamio import NativeBam
    prof = workloads.profile("c2", 0.012)
    b, genome, g_off, meta = workloads.make_batch_full(prof, seed=11, device="cuda:0")
    hb = b.slice_records(0, b.n_rec)
    refs = list(hb.references)
    lens = [int(x) for x in (g_off[1:] - g_off[:-1]).tolist()]
    path = str(tmp_path / "p.bam")
    n_written, raw = harness.write_bam_from_batch(path, hb, refs, lens)
    assert n_written == b.n_rec
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5,
                   "partition_max_distance": 1000, "position_distance_normalizer": 900, "edit_distance_normalizer": 1.0,
                   "cluster_max_distance": 0.5, "all_bnds": True})
    p = _abi.Params.from_options(o)
    e = _lib.Engine(0)
    e.set_genome(g_off, genome, on_device=True)
    # one batch, dense SEQ
    nb = NativeBam(path, threads=4)
    whole, n = nb.read_batch(1 << 30, 20, "coordinate")
    assert n == b.n_rec
    one_sig, one_bnd = e.collect(whole, p)
    crank = batch.contig_ranks(refs)
    one_ct = e.cluster(p, crank, source=0)
    osig, obnd = oracle.collect(whole, p)
    oracle.set_genome(g_off.cpu().numpy().astype(np.int64), genome.cpu().numpy())
    oct_ = oracle.cluster(p, crank, source=0)
    one_names = nb.read_names()
    nb.close()
    assert one_sig.first_difference(osig) is None and one_bnd.first_difference(obnd) is None
    assert one_ct.first_difference(oct_, rtol=1e-12) is None
    # many batches through the pipeline (the device-resident reader: read ids are numbered in another order - compared through the names)
    pipe = harness.BamPipeline(path, o, e, threads=4, batch_records=max(64, n // 7))
    assert pipe.run() == n and pipe.stats["batches"] >= 7
    pipe.cluster()
    many_ct = e.fetch_clusters()
    counts = e.collect_counts()
    many_names = pipe.bam.read_names()
    pipe.close()
    assert e.collect_counts() == counts == (one_sig.n, int(one_sig.seq_off[one_sig.n]), one_bnd.n)     # still resident after accumulation ends
    many_sig, many_bnd = e.fetch_signatures(0), e.fetch_signatures(1)
    for a, c in ((many_sig, one_sig), (many_bnd, one_bnd)):
        for k in _abi.SIG_DTYPES:
            if k not in ("key", "read_id"):
                assert np.array_equal(getattr(a, k)[:a.n], getattr(c, k)[:c.n]), k
        assert [many_names[int(i)] for i in a.read_id[:a.n]] == [one_names[int(i)] for i in c.read_id[:c.n]]
        assert np.all(np.diff(a.key[:a.n].astype(np.int64)) > 0)                                       # one global emission order
        assert np.array_equal(a.seq_off, c.seq_off) and np.array_equal(a.seq[:int(a.seq_off[a.n])], c.seq[:int(c.seq_off[c.n])])
    assert many_ct.first_difference(one_ct) is None
    # CLUSTER again from the promoted lists (what cluster_sv_signatures does after analyze_alignment_file_coordsorted(path))
    again = e.cluster(p, crank, source=0)
    assert again.first_difference(one_ct) is None
    e.close()


def test_long_cigar_cg_tag_on_the_gpu(eng, oracle, tmp_path):
    """> 65535 CIGAR operations (CG:B,I tag): native reader, dense and sparse SEQ -> svx_collect == the oracle."""
    from svim_amd.bamio import NativeBam
    short, a, cig = H.long_cigar_records()
    path = str(tmp_path / "cg.bam")
    records.write_bam(path, ["chr1"], [400000], [short, a])
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5,
                   "partition_max_distance": 1000, "position_distance_normalizer": 900, "edit_distance_normalizer": 1.0,
                   "cluster_max_distance": 0.5, "all_bnds": False})
    p = _abi.Params.from_options(o)
    for filt in (0, 40):
        nb = NativeBam(path, threads=2)
        if filt:
            nb.set_seq_filter(filt)
        b, n = nb.read_batch(1 << 30, 20, "coordinate")
        exp, _ = oracle.collect(b, p)
        got, _ = eng.collect(b, p)
        assert exp.n == 2 and got.first_difference(exp) is None
        assert got.sequence(int(np.nonzero(got.type[:2] == 1)[0][0])) == "ACGT" * 14 + "A"
        nb.close()


def test_g6_distance_bit_patterns_on_the_gpu(eng):
    """Every FP64 span_position_distance the reference returned (tests/golden/g6_distance.json.gz: all six types, insertions with
    real haplotype edit distances) recomputed by the device code of the clustering (svx_pair_distances) - identical bit patterns."""
    g = H.load("g6_distance.json.gz")
    sigs = [H.row_sig(r) for r in g["signatures"]]
    tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(H.REFS))
    off, codes = convert.genome_arrays(H.options({}).genome, contigs.names)
    eng.set_genome(off, codes)
    p = _abi.Params.from_options(H.options({}))
    d = eng.pair_distances(tab, [(i, j) for i, j, _ in g["pairs"]], p)
    bad = [(i, j, hexd, struct.pack("<d", float(x)).hex()) for (i, j, hexd), x in zip(g["pairs"], d) if struct.pack("<d", float(x)).hex() != hexd]
    assert not bad, bad[:5]
    assert len(g["pairs"]) > 500 and {g["signatures"][i][0] for i, _, _ in g["pairs"]} == {"DEL", "INS", "INV", "DUP_TAN", "DUP_INT", "BND"}


def test_per_read_entry_points_match_reference(eng):
    """analyze_alignment_indel / analyze_read_segments (the per-read functions of SVIM_intra.py / SVIM_inter.py) through the
    drop-in names, record by record, against what the reference returned."""
    import svim_amd
    g = H.load("g_entrypoints.json.gz")
    g2 = H.load("g2_collect.json.gz")
    text = [c for c in g2["cases"] if c["name"] == "fuzzA" and c["mode"] == "coordinate" and c.get("sam")][0]["sam"]
    bam = records.AlignmentFile(text=text)
    recs = list(bam.fetch(until_eof=True))
    for run in g["runs"]:
        o = H.options(run["options"])
        for e in run["per_record"][:120]:
            a = recs[e["rec"]]
            s1, b1 = svim_amd.analyze_alignment_indel(a, bam, a.query_name, o)
            assert [H.sig_row(s) for s in s1] == e["indel"], e["rec"]
            assert [H.sig_row(s) for s in b1] == e["indel_bnd"], e["rec"]
            if "segments" in e:
                sup = [x for x in svim_amd.retrieve_other_alignments(a, bam) if x.mapping_quality >= o.min_mapq]
                s2, b2 = svim_amd.analyze_read_segments(a, sup, bam, o)
                assert [H.sig_row(s) for s in s2] == e["segments"], e["rec"]
                assert [H.sig_row(s) for s in b2] == e["segments_bnd"], e["rec"]


def test_batched_per_read_entry_points_match_reference(eng):
    """analyze_alignment_indel_batch / analyze_read_segments_batch: many reads, ONE launch - per record exactly what the reference's per-read
    functions returned (src/svim/SVIM_intra.py:33-51, src/svim/SVIM_inter.py:24-302; g_entrypoints.json.gz)."""
    import svim_amd
    g = H.load("g_entrypoints.json.gz")
    g2 = H.load("g2_collect.json.gz")
    text = [c for c in g2["cases"] if c["name"] == "fuzzA" and c["mode"] == "coordinate" and c.get("sam")][0]["sam"]
    bam = records.AlignmentFile(text=text)
    recs = list(bam.fetch(until_eof=True))
    for run in g["runs"]:
        o = H.options(run["options"])
        per = run["per_record"]
        alns = [recs[e["rec"]] for e in per]
        res = svim_amd.analyze_alignment_indel_batch(alns, bam, [a.query_name for a in alns], o)
        assert len(res) == len(per)
        for e, (s1, b1) in zip(per, res):
            assert [H.sig_row(s) for s in s1] == e["indel"], e["rec"]
            assert [H.sig_row(s) for s in b1] == e["indel_bnd"], e["rec"]
        seg = [e for e in per if "segments" in e]
        reads = []
        for e in seg:
            a = recs[e["rec"]]
            reads.append((a, [x for x in svim_amd.retrieve_other_alignments(a, bam) if x.mapping_quality >= o.min_mapq]))
        res2 = svim_amd.analyze_read_segments_batch(reads, bam, o)
        assert len(res2) == len(seg) and len(seg) > 0
        for e, (s2, b2) in zip(seg, res2):
            assert [H.sig_row(s) for s in s2] == e["segments"], e["rec"]
            assert [H.sig_row(s) for s in b2] == e["segments_bnd"], e["rec"]
    assert svim_amd.analyze_alignment_indel_batch([], bam, [], o) == [] and svim_amd.analyze_read_segments_batch([], bam, o) == []


class _Cand(object):
    """stand-in with the constructor / accessors of svim.SVCandidate.CandidateDuplicationInterspersed"""
    type = "DUP_INT"

    def __init__(self, source_contig, source_start, source_end, dest_contig, dest_start, dest_end, members, score, std_span, std_pos,
                 cutpaste=False):
        self.source_contig, self.source_start, self.source_end = source_contig, max(0, source_start), source_end
        self.dest_contig, self.dest_start, self.dest_end = dest_contig, max(0, dest_start), dest_end
        self.members, self.score, self.std_span, self.std_pos, self.cutpaste = members, score, std_span, std_pos, cutpaste

    def get_source(self):
        return (self.source_contig, self.source_start, self.source_end)

    def get_destination(self):
        return (self.dest_contig, self.dest_start, self.dest_end)

    def get_key(self):
        return (self.type, self.source_contig, self.source_end)

    def downstream_distance_to(self, other):
        if self.type == other.type and self.source_contig == other.source_contig:
            return max(0, other.source_start - self.source_end)
        return float("inf")


def test_partition_and_cluster_candidates_matches_reference(eng):
    import svim_amd
    g = H.load("g_entrypoints.json.gz")
    cands = [_Cand(r[0], r[1], r[2], r[3], r[4], r[5], list(r[10]), r[6], r[7], r[8], r[9]) for r in g["candidates"]]
    res = svim_amd.partition_and_cluster_candidates(cands, H.options(g["options"]), "interspersed duplication candidates")
    got = [[c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end, c.score, c.std_span, c.std_pos,
            bool(c.cutpaste), list(c.members)] for c in res]
    assert len(got) == len(g["merged_candidates"])
    for a, b in zip(got, g["merged_candidates"]):
        assert a[:7] == b[:7] and a[9:] == b[9:], (a, b)
        assert H.close(a[7], b[7]) and H.close(a[8], b[8])


@pytest.mark.parametrize("kmax", ["0", "1", "2"])
def test_edit_distance_row_blocks_vs_oracle(oracle, monkeypatch, kmax):
    """SVX_EDIT_BLOCKED=1 (round 6): the round-0 pairs of the 2-, 4- (and 8-) lane full-matrix families run as row blocks that only walk the columns inside the
    band of the pair's own upper bound, the blocks of a pair in successive launches with their boundary rows handed over through memory.  Pairs at every block-count
    boundary of the four word widths, with a large length gap (what sends a pair to a full matrix in the first place): related ones (one long deletion + a few edits:
    tight upper bound, narrow band, big corners), loosely related and unrelated ones (upper bound = the trivial one: nothing is spared).  Equal to the default route
    on every pair and to the oracle on a sample."""
    from svim_amd._lib import Engine
    rng = random.Random(29)
    pairs = []
    for m in (330, 512, 513, 639, 640, 641, 700, 767, 768, 769, 895, 896, 897, 1000, 1023, 1024, 1025, 1279, 1280, 1281, 1400, 1535, 1536, 1537, 1700, 1791, 1792, 1793,
              1900, 2047, 2048, 2049, 2300, 2559, 2560, 2561, 3000, 3071, 3072, 3073, 3500, 3583, 3584, 3585, 4000, 4095, 4096):
        for gap in (650, 1100, 2500):
            for kind in range(6):
                b = synth.random_seq(rng, m + gap)
                if kind == 3:
                    a = synth.random_seq(rng, m)                                    # unrelated
                elif kind >= 4:
                    # the shorter string is the longer one's head (kind 4: the left-justified alignment bounds the distance tightly - a narrow band, the largest
                    # corners) or its tail (kind 5), with a few substitutions spread over it
                    a = list(b[:m] if kind == 4 else b[gap:])
                    for q in range(7, m, max(1, m // 9)):
                        a[q] = "ACGT"[("ACGT".index(a[q]) + 1) % 4]
                    a = "".join(a)
                else:
                    cut = rng.randrange(0, m)
                    a = list(b[:cut] + b[cut + gap:])                                # one long deletion ...
                    for _ in range((0, 6, m // 12)[kind]):                           # ... and none / a few / many point edits
                        q = rng.randrange(len(a))
                        a[q] = rng.choice("ACGT")
                    a = "".join(a)
                pairs.append((a, b) if rng.random() < 0.5 else (b, a))
    # (a call of at most SVX_EDIT_FEW_PAIRS pairs takes the low-latency route: short filler pairs make this one a large call)
    filler = [(synth.random_seq(rng, 40), synth.random_seq(rng, 44)) for _ in range(2100)]
    idx = rng.sample(range(len(pairs)), 120)
    exp = {i: oracle.edit_distance(*pairs[i]) for i in idx}
    results = []
    # (SVX_EDIT_BLOCKED_WALK: pairs that walk more columns per block stay in the multi-lane forms - the default split in one variant, everything in row blocks in the others)
    walk = {} if kmax == "2" else {"SVX_EDIT_BLOCKED_WALK": "1000000"}
    for env in ({}, dict({"SVX_EDIT_BLOCKED": "1", "SVX_EDIT_BLOCKED_K": kmax}, **walk)):
        for k in ("SVX_EDIT_BLOCKED", "SVX_EDIT_BLOCKED_K", "SVX_EDIT_BLOCKED_WALK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = Engine()
        try:
            results.append(e.edit_distances(pairs + filler))
        finally:
            e.close()
    assert results[0] == results[1]
    for i in idx:
        assert results[1][i] == exp[i], (i, len(pairs[i][0]), len(pairs[i][1]))


@pytest.mark.parametrize("few_pairs", [None, "0"])
def test_edit_distance_full_matrix_classes_vs_oracle(oracle, monkeypatch, few_pairs):
    """Unrelated pairs at every row-count boundary of the full-matrix classes (one lane <= 512 rows; 2/4/8/16 lanes of 10, 12, 14 or 16
    words; systolic beyond 8192), short and long texts.  A call this small takes the low-latency route (64-lane forms) by default;
    SVX_EDIT_FEW_PAIRS=0 sends the same pairs through the classes a large call uses."""
    from svim_amd._lib import Engine
    if few_pairs is None:
        monkeypatch.delenv("SVX_EDIT_FEW_PAIRS", raising=False)
    else:
        monkeypatch.setenv("SVX_EDIT_FEW_PAIRS", few_pairs)
    rng = random.Random(17)
    pairs = []
    for m in (1, 31, 32, 33, 64, 65, 128, 129, 256, 257, 511, 512, 513, 640, 641, 700, 768, 769, 896, 897, 1000, 1024, 1025, 1280, 1281, 1500, 1536, 1537,
              1792, 1793, 2048, 2049, 2560, 2561, 3000, 3072, 3073, 3584, 3585, 4096, 4097, 5120, 5121, 6000, 6144, 6145, 7168, 7169, 8192, 8193):
        for extra in (0, 3, 700):
            a = synth.random_seq(rng, m)
            b = synth.random_seq(rng, m + extra)
            pairs.append((a, b) if rng.random() < 0.5 else (b, a))
    e = Engine()
    try:
        got = e.edit_distances(pairs)
    finally:
        e.close()
    exp = [oracle.edit_distance(a, b) for a, b in pairs]
    assert got == exp


_NARROW_CACHE = {}


@pytest.mark.parametrize("guess", ["pilot", "0.03", "0.12", "0.3"])
def test_edit_distance_narrowing_windows_vs_oracle(oracle, monkeypatch, guess):
    """The staircase windows narrow while they run (d_edit_stair: Ukkonen cut-off per 32-column block, wave-uniform).  Pairs built to sit where a
    wrong cut would show: the cheap alignment runs tens of diagonals away from the corridor for most of its length (a deletion near the start
    paid back by an insertion near the end, either side), differences bunched at the start / at the end / in the middle, long clean stretches
    (nothing may be cut there), noise as substitutions only and as indels, equal and unequal lengths; enough pairs that whole waves share a class,
    and ragged lengths so that lanes of one wave finish in different blocks.  Every distance must be the oracle's, whatever the speculation."""
    from svim_amd._lib import Engine
    monkeypatch.delenv("SVX_EDIT_NARROW", raising=False)
    monkeypatch.delenv("SVX_EDIT_FORCE_FULL", raising=False)
    if guess == "pilot":
        monkeypatch.delenv("SVX_EDIT_GUESS", raising=False)
    else:
        monkeypatch.setenv("SVX_EDIT_GUESS", guess)
    monkeypatch.setenv("SVX_EDIT_FEW_PAIRS", "0")
    rng = random.Random(29)

    def noisy(s, rate, lo, hi, indels):
        b = list(s)
        k = int(rate * (hi - lo))
        for _ in range(k):
            p = rng.randrange(lo, max(lo + 1, min(hi, len(b))))
            r = rng.random() if indels else 0.0
            if r < 0.5:
                b[p] = rng.choice("ACGT")
            elif r < 0.75:
                del b[p]
            else:
                b.insert(p, rng.choice("ACGT"))
        return "".join(b)

    pairs = []
    for it in range(4600):
        la = rng.choice((900, 1700, 2500, 3300, 4300, 5200)) + rng.randrange(0, 90)
        a = synth.random_seq(rng, la)
        kind = it % 8
        rate = rng.choice((0.01, 0.03, 0.05, 0.08))
        indels = rng.random() < 0.5
        if kind == 0:                                   # uniform noise
            b = noisy(a, rate, 0, la, indels)
        elif kind == 1:                                 # all differences in the last fifth
            b = noisy(a, 4 * rate, la - la // 5, la, indels)
        elif kind == 2:                                 # all in the first fifth
            b = noisy(a, 4 * rate, 0, la // 5, indels)
        elif kind == 3:                                 # a block missing near the start, another one inserted near the end: the path runs off the corridor in between
            k = rng.choice((10, 25, 45, 70))
            b = noisy(a, rate * 0.5, 0, la, indels)
            b = b[:30] + b[30 + k:]
            b = b[:len(b) - 40] + synth.random_seq(rng, k) + b[len(b) - 40:]
        elif kind == 4:                                 # the other side
            k = rng.choice((10, 25, 45, 70))
            b = noisy(a, rate * 0.5, 0, la, indels)
            b = b[:30] + synth.random_seq(rng, k) + b[30:]
            b = b[:len(b) - 40 - k] + b[len(b) - 40:]
        elif kind == 5:                                 # clean halves around a noisy middle
            b = noisy(a, 6 * rate, 2 * la // 5, 3 * la // 5, indels)
        elif kind == 6:                                 # unequal lengths: the corridor itself is wide
            b = noisy(a, rate, 0, la, indels) + synth.random_seq(rng, rng.choice((20, 60, 150)))
        else:                                           # shifted copy (what insertion pairs at different positions look like)
            k = rng.choice((5, 20, 40))
            b = synth.random_seq(rng, k) + noisy(a, rate, 0, la, indels)[:la - k]
        if rng.random() < 0.5:
            a, b = b, a
        pairs.append((a, b))
    e = Engine()
    try:
        got = e.edit_distances(pairs)
        st = e.stats()
    finally:
        e.close()
    idx = range(len(pairs))
    if "exp" not in _NARROW_CACHE:
        _NARROW_CACHE["exp"] = {i: oracle.edit_distance(*pairs[i]) for i in idx}          # the pairs do not depend on the parameter
    exp = _NARROW_CACHE["exp"]
    bad = [(i, got[i], exp[i], len(pairs[i][0]), len(pairs[i][1]), i % 8) for i in idx if got[i] != exp[i]]
    assert not bad, bad[:10]
    assert st["n_edit_wordcols_band"] > 0
    # the switch only changes the work, never a result
    monkeypatch.setenv("SVX_EDIT_NARROW", "0")
    e = Engine()
    try:
        wide = e.edit_distances(pairs)
        st0 = e.stats()
    finally:
        e.close()
    assert wide == got
    assert st["n_edit_wordcols_band"] < st0["n_edit_wordcols_band"]          # narrowing did happen


def test_edit_distance_routes_do_not_change_results(oracle, monkeypatch):
    """The band speculation (SVX_EDIT_GUESS pinned tiny / huge, or chosen by the per-call pilot from a sample of the call's own
    pairs), the forced full-matrix route and the route of a small retry round are performance choices only: every route must return the
    oracle's distances."""
    from svim_amd._lib import Engine
    rng = random.Random(5)
    pairs = []
    for _ in range(5000):
        la = rng.choice((150, 700, 1400, 2600, 4500))
        a = synth.random_seq(rng, la)
        b = list(a)
        for _ in range(int(rng.choice((0.0, 0.02, 0.06, 0.15)) * la)):
            p = rng.randrange(len(b))
            r = rng.random()
            if r < 0.4:
                b[p] = rng.choice("ACGT")
            elif r < 0.7:
                del b[p]
            else:
                b.insert(p, rng.choice("ACGT"))
        b = "".join(b)
        if rng.random() < 0.2:
            b = synth.random_seq(rng, rng.choice((9, 60))) + b               # shifted: the Hamming bound is useless, the band is a guess
        if rng.random() < 0.05:
            b = synth.random_seq(rng, rng.randrange(50, 3000))               # unrelated
        pairs.append((a, b))
    exp = [oracle.edit_distance(a, b) for a, b in pairs[:400]]
    results = []
    for env in ({"SVX_EDIT_GUESS": "0.004"}, {"SVX_EDIT_GUESS": "0.45"}, {"SVX_EDIT_FORCE_FULL": "1"}, {"SVX_EDIT_NARROW": "0"},
                {"SVX_EDIT_NARROW": "0", "SVX_EDIT_GUESS": "0.2"}, {"SVX_EDIT_GUESS": "0.2"},
                # retry rounds: band pairs of a small retry round as systolic full matrices (round 6) - never / always / default (<= 4096 pairs);
                # the host's waits through the mailbox or through copy + stream synchronisation
                {"SVX_EDIT_GUESS": "0.004", "SVX_EDIT_RETRY_FULL": "0"}, {"SVX_EDIT_GUESS": "0.004", "SVX_EDIT_RETRY_FULL": "1000000"},
                {"SVX_EDIT_RETRY_FULL": "0"}, {"SVX_MAILBOX": "0"}, {}):
        for k in ("SVX_EDIT_GUESS", "SVX_EDIT_FORCE_FULL", "SVX_EDIT_NARROW", "SVX_EDIT_RETRY_FULL", "SVX_MAILBOX"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = Engine()
        try:
            first = e.edit_distances(pairs)
            second = e.edit_distances(pairs)       # a context carries no speculation state from call to call
        finally:
            e.close()
        assert first == second
        results.append(first)
    assert all(r == results[0] for r in results)
    assert results[0][:400] == exp


@pytest.mark.parametrize("env, normalizer", [({}, 900), ({"SVX_EDIT_NO_PREPACK": "1"}, 900), ({"SVX_EDIT_NO_EARLY": "1"}, 900), ({}, 40000), ({}, 5)])
def test_cluster_scheduling_switches_do_not_change_results(oracle, monkeypatch, env, normalizer):
    """The haplotype store packed ahead of the pair list (radius from the parameters; off when 2 * cluster_max_distance * normalizer is out of
    range: normalizer 40000 -> the exact radius comes from the pair list) and the early full-matrix retries are scheduling choices: the tables
    stay those of the oracle."""
    from svim_amd._lib import Engine
    for k in ("SVX_EDIT_NO_PREPACK", "SVX_EDIT_NO_EARLY"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    bam, refs, references = _planted_case(91, 3000, 80)
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10,
                   "segment_overlap_tolerance": 5, "partition_max_distance": 1000, "position_distance_normalizer": normalizer,
                   "edit_distance_normalizer": 1.0, "cluster_max_distance": 0.5, "all_bnds": False})
    hb = batch.build_batch(bam, o, mode="coordinate")
    p = _abi.Params.from_options(o)
    off, codes = convert.genome_arrays(refs, references)
    e = Engine()
    try:
        e.set_genome(off, codes)
        oracle.set_genome(off, codes)
        sig, bnd = e.collect(hb, p)
        osig, obnd = oracle.collect(hb, p)
        assert sig.first_difference(osig) is None
        ct = e.cluster(p, hb.contig_rank, source=0)
        oc = oracle.cluster(p, hb.contig_rank, source=0)
        assert ct.type_count[_abi.SVX_INS] > 0 and e.stats()["n_edit_pairs"] > 0
        assert ct.first_difference(oc, rtol=1e-12) is None
    finally:
        e.close()


def test_bench_harness_on_a_bam_and_fasta(tmp_path):
    """`python bench.py --bam X --fasta Y` (the harness for real inputs): one JSON line on stdout and nothing else, counts equal to
    the reference's configs[0] result."""
    import json
    import os
    import subprocess
    import sys
    g, refs, recs = H.c1_case()
    path = str(tmp_path / "c1.bam")
    records.write_bam(path, ["chr1"], [2000000], recs)
    fa = str(tmp_path / "c1.fa")
    synth.write_fasta(fa, refs)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--bam", path, "--fasta", fa, "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["unit"] == "reads/s" and d["value"] > 0 and d["data"] == "file"
    assert d["counts"]["records"] == len(recs)
    assert d["counts"]["signatures"] == len(g["signatures"])
    assert d["counts"]["clusters"] == sum(len(x) for x in g["clusters"])
    assert d["end_to_end"]["bam_file_reads_per_s"] == d["value"]


def test_bench_under_torchrun_multi_gpu_code_path_single_rank():
    """The launch the driver uses for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), here with one rank and the multi-GPU code path
    forced (RCCL process group from torchrun's environment, contig-sharded step, gathers): ONE JSON line on stdout, RCCL's banner and
    everything else on stderr."""
    import json
    import os
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env["SVX_BENCH_FORCE_DIST"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--reads", "100000", "--contig-len", "25000000", "--no-cpu-baseline", "--no-end-to-end"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["counts"]["signatures"] > 10000 and d["counts"]["clusters"] > 1000


@pytest.mark.parametrize("decoder", ["wave_per_block", "lane_per_block"])
def test_bgzf_inflate_on_the_gpu_equals_zlib(tmp_path, monkeypatch, decoder):
    """svx_inflater returns zlib's bytes: the blocks of a BAM the writer made at its usual level, the same records at level 9 and level 1, stored blocks
    (level 0), an empty block (the EOF marker) and incompressible data - with the default decoder (one wavefront per BGZF block, svim_amd/csrc/inflate_core.hpp)
    and with the opt-in one (SVX_INFLATE_LANES=1: one LANE per block, inflate_lanes.hpp; the blocks a lane gives up - stored ones, code tables beyond its
    share of LDS - are redone by the first, so the result must be the same bytes; a damaged block must still be reported)."""
    import gzip
    import zlib
    from svim_amd._lib import Inflater, bgzf_blocks
    contigs = [("chr1", 150000)]
    refs = synth.make_reference(3, contigs)
    recs = synth.planted_reads(5, 500, refs, ["chr1"], [150000], n_sites=25, types=("DEL", "INS", "INV"))
    path = str(tmp_path / "t.bam")
    records.write_bam(path, ["chr1"], [150000], synth.coordinate_sort(recs))
    blocks = bgzf_blocks(path)
    assert len(blocks) > 20 and blocks[-1][1] == 0                               # the BGZF EOF marker is an empty block
    with gzip.open(path, "rb") as fh:
        whole = fh.read()
    # the same content re-deflated at other levels, in 60 000-byte blocks, plus random bytes (stored / barely compressible)
    rng = random.Random(3)
    noise = bytes(rng.getrandbits(8) for _ in range(50000))
    for level in (0, 1, 9):
        for lo in range(0, min(len(whole), 600000), 60000):
            raw = whole[lo:lo + 60000]
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            blocks.append((co.compress(raw) + co.flush(), len(raw)))
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        blocks.append((co.compress(noise) + co.flush(), len(noise)))
    if decoder == "lane_per_block":
        monkeypatch.setenv("SVX_INFLATE_LANES", "1")
    else:
        monkeypatch.delenv("SVX_INFLATE_LANES", raising=False)
    f = Inflater(0)
    try:
        got = f.inflate(blocks)
        expect = b"".join(zlib.decompress(b, -15) if s else b"" for b, s in blocks)
        assert len(got) == len(expect) and got.tobytes() == expect
        # a block with a damaged payload among sound ones: refused by either decoder (the lane gives it up, the wave-per-block decoder reports it)
        bad = blocks[3][0][:len(blocks[3][0]) // 2]                                # cut in the middle: it cannot produce its ISIZE
        with pytest.raises(Exception):
            f.inflate(blocks[:3] + [(bytes(bad), blocks[3][1])] + blocks[4:8])
        got = f.inflate(blocks[:8])                                              # and the inflater still works afterwards
        assert got.tobytes() == b"".join(zlib.decompress(b, -15) if s else b"" for b, s in blocks[:8])
    finally:
        f.close()


@pytest.mark.parametrize("sub", ["16", "40", None])
def test_reader_with_gpu_inflate_delivers_the_same_batches(tmp_path, monkeypatch, sub):
    """svx_bam_set_gpu_inflate: the GPU inflates sub-batches from the front of every chunk, the host's cores blocks from its back (sub-batches of
    16 / 40 blocks here so that both sides work on one small file and the three slots rotate; None = the default sizing).  Every array of every
    batch equals what the host-only reader delivers, in both SEQ modes."""
    from svim_amd.bamio import NativeBam
    if sub is None:
        monkeypatch.delenv("SVX_BAM_GPU_SUB", raising=False)
    else:
        monkeypatch.setenv("SVX_BAM_GPU_SUB", sub)
    monkeypatch.setenv("SVX_BAM_CHUNK_BLOCKS", "120")                    # several chunks: the window switch happens with page-locked buffers
    contigs = [("chr1", 200000), ("chr2", 80000)]
    refs = synth.make_reference(3, contigs)
    references, lengths = [c[0] for c in contigs], [c[1] for c in contigs]
    recs = synth.planted_reads(5, 1500, refs, references, lengths, n_sites=60, types=("DEL", "INS", "INV"))
    recs += synth.fuzz_split_reads(6, 150, references, lengths)
    path = str(tmp_path / "t.bam")
    records.write_bam(path, references, lengths, synth.coordinate_sort(recs))

    def read_all(gpu, seq_filter):
        nb = NativeBam(path, threads=4)
        if seq_filter:
            nb.set_seq_filter(40)
        if gpu:
            nb.set_gpu_inflate(0)
        out = []
        for _ in range(2):                                                 # second pass after rewind: buffers reused
            while True:
                b, k = nb.read_batch(700, 20, "coordinate")
                if k == 0:
                    break
                out.append(nb.batch_arrays(b))
            nb.rewind()
        st = nb.gpu_inflate_stats()
        nb.close()
        return out, st

    for seq_filter in (False, True):
        host, _ = read_all(False, seq_filter)
        both, st = read_all(True, seq_filter)
        assert st["gpu_blocks"] > 0 and (sub is None or st["cpu_blocks"] > 0), st
        assert len(host) == len(both) and len(host) > 4
        for a, b in zip(host, both):
            assert a.keys() == b.keys()
            for k in a:
                assert np.array_equal(a[k], b[k]), k


def test_combine_consumers_on_gpu_cluster_lists(eng):
    """SURVEY 8f row 4 on the GPU: cluster_sv_signatures (drop-in name -> svx_cluster) hands out lazy ClusterLists; the COMBINE-side consumers
    (src/svim/SVIM_merging.py:12-29,93-159, SVIM_COMBINE.py:332-478 - replayed by tests/combine_consumer.py) delete from / extend them and must
    arrive at the candidates the reference's own functions computed (tests/golden/g_combine.json.gz)."""
    import combine_consumer as cc
    import svim_amd
    from svim_amd.lazy import ClusterList
    g = H.load("g_combine.json.gz")
    o = H.options(g["options"])
    sigs = [H.row_sig(r) for r in g["signatures"]]
    clusters = svim_amd.cluster_sv_signatures(sigs, o)
    assert all(isinstance(c, ClusterList) and c._objs is None for c in clusters)
    idx = {id(s): i for i, s in enumerate(sigs)}
    got_rows = []
    for k, lst in enumerate(clusters):
        rows = []
        for c in lst:
            mem = [idx[id(m)] for m in c.members]
            if k < 3:
                rows.append([c.contig, c.start, c.end, c.score, c.size, c.std_span, c.std_pos, mem])
            else:
                rows.append([c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end, c.score, c.size, c.std_span, c.std_pos, mem]
                            + ([c.direction1, c.direction2] if c.type == "BND" else []))
        got_rows.append(rows)
    H.compare_cluster_rows(got_rows, g["clusters"])
    got = cc.consume(clusters, o, idx)
    diff = H.first_json_difference(got, g["expected"])
    assert diff is None, diff
    assert len(clusters[1]) == g["expected"]["n_ins_after"] and len(clusters[5]) == g["expected"]["n_bnd_after_merge"]


def test_writers_on_gpu_fetched_clusters(eng, tmp_path):
    """write_signature_clusters_bed / _vcf (src/svim/SVIM_CLUSTER.py:29-106) on clusters FETCHED FROM THE GPU (the drop-in cluster_sv_signatures) ==
    the files the reference's writers produced for the reference's clusters (tests/golden/g_writers.json.gz); FP columns within 1e-9."""
    import os
    import svim_amd
    from svim_amd import SVIM_CLUSTER
    gold = H.load("g_writers.json.gz")
    g5 = H.load("g5_cluster.json.gz")
    case = [c for c in g5["cases"] if c["name"] == "stress31"][0]
    sigs = [H.row_sig(r) for r in case["signatures"]]
    clusters = svim_amd.cluster_sv_signatures(sigs, H.options(case["options"]))
    SVIM_CLUSTER.write_signature_clusters_bed(str(tmp_path), clusters)
    SVIM_CLUSTER.write_signature_clusters_vcf(str(tmp_path), clusters, gold["version"])
    for name, exp in gold["files"].items():
        with open(os.path.join(str(tmp_path), name)) as fh:
            got = fh.read()
        assert H.text_close(got, exp) is None, (name, H.text_close(got, exp))
    # and after a COLLECT -> CLUSTER that stayed resident (lazy SignatureList in, lazy ClusterLists out): the writers read members lazily
    g = H.load("g2_collect.json.gz")
    ccase = [c for c in g["cases"] if c["name"] == "fuzzA" and c["mode"] == "coordinate" and not c["options"]["all_bnds"]][0]
    o = H.options(ccase["options"])
    lazy_sigs, _ = svim_amd.analyze_alignment_file_coordsorted(records.AlignmentFile(text=ccase["sam"]), o)
    res = svim_amd.cluster_sv_signatures(lazy_sigs, o)
    d2 = tmp_path / "resident"
    d2.mkdir()
    SVIM_CLUSTER.write_signature_clusters_bed(str(d2), res)
    SVIM_CLUSTER.write_signature_clusters_vcf(str(d2), res, gold["version"])
    listed = sorted(os.path.relpath(os.path.join(r, f), str(d2)) for r, _, fs in os.walk(str(d2)) for f in fs)
    assert listed == sorted(gold["files"])
    n_lines = 0
    for name in listed:
        with open(os.path.join(str(d2), name)) as fh:
            n_lines += sum(1 for line in fh if not line.startswith("#"))
    assert n_lines >= sum(len(x) for x in res)


def test_c1_config0_full_size_through_bam_file(eng, tmp_path):
    """BASELINE.json configs[0] AT ITS STATED SIZE (250 Mb contig, 10 000 reads ~ triangular(100, 20000, 15000), 1.2*10^7 CIGAR
    operations) end to end through the drop-in entry points: BAM file + FASTA on disk -> native reader (GPU inflate) -> analyze_alignment_file_coordsorted
    -> cluster_sv_signatures == what the reference's CPU path returned (tests/golden/g_c1_full.json.gz)."""
    import svim_amd
    from svim_amd import harness
    g = H.load("g_c1_full.json.gz")
    hb, genome, meta = H.c1_full_case()
    assert hb.n_rec == g["n_records"] and int(meta["n_ops"]) == g["n_ops"]
    path, fa = str(tmp_path / "c1_full.bam"), str(tmp_path / "c1_full.fa")
    harness.write_bam_from_batch(path, hb, ["chr1"], [int(genome.shape[0])])
    H.write_fasta_from_codes(fa, "chr1", genome)
    o = H.options(g["options"])
    o.genome = fa
    sigs, bnds = svim_amd.analyze_alignment_file_coordsorted(path, o)
    assert [H.sig_row(s) for s in sigs] == g["signatures"] and len(bnds) == g["n_bnds"]
    res = svim_amd.cluster_sv_signatures(sigs, o)
    idx = {id(s): i for i, s in enumerate(sigs)}
    got = [[[c.contig, c.start, c.end, c.score, c.size, c.std_span, c.std_pos, [idx[id(m)] for m in c.members]] for c in lst] if k < 3 else [] for k, lst in enumerate(res)]
    assert all(len(lst) == 0 for lst in res[3:])
    H.compare_cluster_rows(got, g["clusters"])


def _large_partition_table(drop_set_method):
    """g5's 'stress31' signature list (3 contigs, 35 partitions beyond 100 members, all six types; three DEL partitions of 1045 / 1046 / 3000 members);
    drop_set_method: without the two partitions beyond 1045 members (random.sample's set method)"""
    g5 = H.load("g5_cluster.json.gz")
    case = [c for c in g5["cases"] if c["name"] == "stress31"][0]
    rows = case["signatures"]
    if drop_set_method:
        rows = [r for r in rows if not (r[0] == "DEL" and r[1] == g5["references"][0] and (39000 <= r[2] <= 41500 or 59000 <= r[2] <= 61500))]
    sigs = [H.row_sig(r) for r in rows]
    tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(g5["references"]))
    return g5, H.options(case["options"]), tab, contigs


def _split_by_owner(tab, owner):
    """per-rank tables holding the rows whose partition key contig belongs to the rank (no foreign rows), emission order kept"""
    from svim_amd import multigpu
    n = tab.n
    own = owner[multigpu.owner_contig(tab.type[:n], tab.contig[:n], np.where(tab.contig2[:n] >= 0, tab.contig2[:n], tab.contig[:n]))]
    out = []
    for r in range(int(owner.max()) + 1):
        idx = np.nonzero(own == r)[0]
        ln = tab.seq_off[idx + 1] - tab.seq_off[idx]
        t = _abi.SigTable(len(idx), int(ln.sum()))
        for k in _abi.SIG_DTYPES:
            getattr(t, k)[:] = getattr(tab, k)[idx]
        t.seq_off[1:] = np.cumsum(ln)
        pos = 0
        for i, l in zip(idx, ln):
            t.seq[pos:pos + l] = tab.seq[tab.seq_off[i]:tab.seq_off[i] + l]
            pos += int(l)
        out.append(t)
    return out


@pytest.mark.parametrize("mode", ["transfer tables", "forced exact rounds", "set-method partitions"])
def test_rank_exchange_three_contexts_on_one_gpu(oracle, monkeypatch, mode):
    """svx_cluster_set_ranks: three contexts on cuda:0, one thread each, stand for three contig-sharded ranks; the all-gather
    transport is an in-process barrier.  Every rank must find - with all-gathers only - the stream positions the serial order implies
    (src/svim/SVIM_clustering.py:129-134), i.e. produce the clusters the oracle produces when it is TOLD those positions (computed here with CPython's
    own generator), and all ranks together the single-process result.  Three protocol paths: composed transfer tables (no partition beyond 1045
    members), the exact rank-after-rank rounds forced on the same input, and the rounds chosen because set-method partitions exist.  A fourth rank
    without signatures takes part with an empty table."""
    import threading
    from svim_amd import _lib, multigpu
    g5, o, tab, contigs = _large_partition_table(drop_set_method=(mode != "set-method partitions"))
    if mode == "forced exact rounds":
        monkeypatch.setenv("SVX_RANKS_EXACT", "1")
    world = 4
    p = _abi.Params.from_options(o)
    crank = batch.contig_ranks(contigs.names)
    off, codes = convert.genome_arrays(o.genome, contigs.names)
    owner = np.asarray([0, 2, 1], dtype=np.int32)                          # name order chr1 < chr10 < chr2: ranks own consecutive ranges of it; rank 3 owns nothing
    assert [contigs.names[i] for i in np.argsort(crank)] == sorted(contigs.names) and list(owner[np.argsort(crank)]) == [0, 1, 2]
    locals_ = _split_by_owner(tab, owner) + [_abi.SigTable(0, 0)]
    ag = H.ThreadAllGather(world)
    results, errors = [None] * world, []

    def work(r):
        try:
            eng = _lib.Engine(0)
            eng.set_genome(off, codes)
            eng.set_ranks(r, world, ag.for_rank(r))
            ct = eng.cluster(p, crank, table=locals_[r])
            results[r] = (ct, eng.stream_positions())
            eng.set_ranks(0, 1, None)
            eng.close()
        except BaseException as e:                                         # noqa: B902
            errors.append((r, e))
            ag.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    # expected: the oracle on every rank's table, told the start positions the serial order implies
    oracle.set_genome(off, codes)
    sizes = []
    for r in range(world):
        per_type = [[] for _ in range(6)]
        t = locals_[r]
        if t.n:
            sidx, pid = oracle.form_partitions(t, crank, int(p.partition_max_distance))
            typ = t.type[:t.n][sidx]
            cut = np.nonzero(np.diff(pid))[0] + 1
            for a, b in zip(np.concatenate([[0], cut]), np.concatenate([cut, [t.n]])):
                if b - a > 100:
                    per_type[int(typ[a])].append(int(b - a))
        sizes.append(per_type)
    assert sum(len(x) for per_type in sizes for x in per_type) >= 25 and sum(1 for per_type in sizes if any(per_type)) >= 3
    total = 0
    for r in range(world):
        starts = [multigpu.stream_words_after([n for q in range(r) for n in sizes[q][t]]) for t in range(6)]
        log = []
        oracle.set_chain(lambda phase, w, starts=starts: w.__setitem__(slice(None), starts) if phase == 0 else log.append(list(w)))
        oc = oracle.cluster(p, crank, table=locals_[r])
        oracle.set_chain(None)
        ct, (got_start, got_end) = results[r]
        assert got_start == starts, (r, got_start, starts)
        assert got_end == (log[0] if log else starts), (r, got_end, log)
        assert ct.first_difference(oc, rtol=1e-12) is None, r
        total += ct.n
    # and all ranks together: the single-process result (no partition spans ranks)
    full = oracle.cluster(p, crank, table=tab)
    assert total == full.n
    assert ag.calls >= 2 * world


def test_two_ranks_share_one_gpu_over_gloo():
    """N > 1 on hardware within the one-GPU box: two processes (torch.distributed.run, backend gloo - RCCL refuses two ranks on one device) both on
    cuda:0 run the contig-sharded step with the real engine and device tensors: foreign BND rows cross the ranks, svx_cluster's rank exchange
    (svx_cluster_set_ranks) runs over the process group, rank 0 gathers; checked against the oracle on the union of both inputs
    (tests/mp_two_ranks_one_gpu.py)."""
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(here, "mp_two_ranks_one_gpu.py")],
                         capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here))
    lines = [l for l in out.stdout.splitlines() if l.startswith("TWO_RANKS_")]
    assert out.returncode == 0 and lines, (out.stdout[-1500:], out.stderr[-3000:])
    assert lines[-1].startswith("TWO_RANKS_OK"), lines[-1]
    _, n_clusters, n_cross, n_foreign = lines[-1].split()
    assert int(n_clusters) > 100 and int(n_cross) > 5 and int(n_foreign) > 0


def test_bench_two_ranks_one_gpu_with_foreign_rows():
    """bench.py --gpus 2 --workload c2 --foreign-frac 0.5 as the driver launches it, but with both ranks on cuda:0 over gloo (SVX_BENCH_BACKEND): the
    timed step carries foreign rows and the rank exchange; ONE JSON line."""
    import json
    import os
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SVX_BENCH_BACKEND="gloo", SVX_BENCH_ONE_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--workload", "c2", "--scale", "0.02", "--foreign-frac", "0.5", "--partition-max-distance", "5000"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["multi_gpu"]["foreign_segments_planted_rank0"] > 0
    assert d["multi_gpu"]["clusters_gathered"] > 100 and len(d["multi_gpu"]["signatures_per_rank"]) == 2


def _run_bench(args, world, backend_env):
    import json
    import os
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **backend_env)
    if world == 1:
        cmd = [sys.executable, os.path.join(repo, "bench.py")] + args
    else:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(repo, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_strong_scaling_two_ranks_equal_one_rank_on_one_gpu():
    """bench.py --scaling strong: ONE whole-genome batch sharded by contig ownership - two ranks (both on cuda:0 over gloo here) process the
    same records, signatures and clusters as one rank; the line names the process group's world size and what crossed the fabric in the last step.  No scaling
    claim: the ranks share a GPU."""
    common = ["--steps", "1", "--warmup", "1", "--scaling", "strong", "--scale", "0.01"]
    one = _run_bench(["--gpus", "1"] + common, 1, {})
    two = _run_bench(["--gpus", "2"] + common, 2, {"SVX_BENCH_BACKEND": "gloo", "SVX_BENCH_ONE_GPU": "1"})
    assert one["scaling"] == two["scaling"] == "strong" and one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["counts"]["reads_used"] > 2000 and one["counts"]["clusters_gathered"] > 100
    for k in ("reads_used", "signatures", "cigar_ops", "clusters_gathered"):
        assert one["counts"][k] == two["counts"][k], (k, one["counts"], two["counts"])
    assert len(two["per_rank"]) == 2 and all(r["records"] > 0 for r in two["per_rank"]) and sum(r["records"] for r in two["per_rank"]) == one["per_rank"][0]["records"]
    mg = two["multi_gpu"]
    assert mg["world_size_seen_by_the_process_group"] == 2 and mg["fabric_last_step_rank0"]["collectives"] >= 4 and mg["fabric_last_step_rank0"]["payload_bytes"] > 0
    assert any("gather(final" in k for k in mg["fabric_last_step_rank0"]["by_kind"])


def test_bench_strong_scaling_over_rccl_when_two_gpus_are_visible():
    """the same over RCCL, one rank per GPU - only where the box has two GPUs (the builder's and the driver's test boxes have one: skipped there)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL refuses two ranks on one device")
    common = ["--steps", "1", "--warmup", "1", "--scaling", "strong", "--scale", "0.01"]
    one = _run_bench(["--gpus", "1"] + common, 1, {})
    two = _run_bench(["--gpus", "2"] + common, 2, {})
    for k in ("reads_used", "signatures", "cigar_ops", "clusters_gathered"):
        assert one["counts"][k] == two["counts"][k], (k, one["counts"], two["counts"])
    assert two["multi_gpu"]["backend"].startswith("nccl") and two["multi_gpu"]["world_size_seen_by_the_process_group"] == 2


def _read_all_batches(nb, batch_records, regions=None):
    """every batch of a reader (host or device-resident) as host arrays + read names per record"""
    out, names_per_rec = [], []
    for region in (regions or [None]):
        if region is not None:
            nb.seek(region[0], region[1])
        while True:
            b, n = nb.read_batch(batch_records, 20, "coordinate")
            if n == 0:
                break
            A = nb.batch_arrays(b)
            names = nb.read_names()
            names_per_rec += [names[int(i)] for i in A["read_id"]]
            out.append(A)
    return out, names_per_rec


def _concat_batches(batches):
    """record-level view that does not depend on how the records were cut into batches: per-record CIGARs, bases, segment rows"""
    rows = []
    for A in batches:
        n = len(A["flag"])
        co, so, sg, sc = A["cigar_off"].astype(np.int64), A["seq_off"].astype(np.int64), A["seg_off"].astype(np.int64), A["seg_cigar_off"].astype(np.int64)
        for i in range(n):
            nbytes = (int(A["lseq"][i]) + 1) // 2
            segs = []
            for r in range(int(sg[i]), int(sg[i + 1])):
                segs.append((int(A["seg_tid"][r]), int(A["seg_pos"][r]), int(A["seg_rev"][r]), int(A["seg_mapq"][r]), int(A["seg_lseq"][r]),
                             A["seg_cigar"][sc[r]:sc[r + 1]].tobytes()))
            rows.append((int(A["flag"][i]), int(A["tid"][i]), int(A["pos"][i]), int(A["mapq"][i]), int(A["lseq"][i]), A["cigar"][co[i]:co[i + 1]].tobytes(),
                         A["seq"][so[i]:so[i] + nbytes].tobytes(), tuple(segs)))
        # emission slots are per batch (the caller adds its slot base): 2 i for the record's indels, 2 i + 1 for its read's split-alignment signatures
        assert np.array_equal(A["order"], 2 * np.arange(n, dtype=np.uint32)) and np.array_equal(A["seg_order"], 2 * np.arange(n, dtype=np.uint32) + 1)
    return rows


@pytest.mark.parametrize("chunk_blocks", [None, "1", "3"])
def test_device_bam_decode_equals_host_reader(tmp_path, monkeypatch, chunk_blocks):
    """BGZF inflate + record discovery + field / CIGAR / SA / name decode ON THE GPU (svx_bam_set_device_decode, csrc/bamdev.hip) hands
    out the records the host reader (bamio.cpp) decodes - flag, tid, pos, mapq, l_seq, packed CIGAR, packed bases, the SA-derived segment table, read names -
    on configs[0] (small geometry), a multi-contig split-read file with SA tags to known and unknown contigs, and a record whose 67 000-operation CIGAR
    lives in the CG tag; also with chunks of one / three BGZF blocks (records and even record HEADERS straddle chunk boundaries)."""
    from svim_amd.bamio import NativeBam
    if chunk_blocks:
        monkeypatch.setenv("SVX_BAM_DEV_CHUNK_BLOCKS", chunk_blocks)
    files = []
    g, refs, recs = H.c1_case()
    p1 = str(tmp_path / "c1.bam")
    records.write_bam(p1, ["chr1"], [2000000], recs[:3000])
    files.append((p1, 777))
    refs3, lens3 = ["chr1", "chr2", "chr10"], [180000, 60000, 60000]
    ref = synth.make_reference(21, list(zip(refs3, lens3)))
    rr = synth.coordinate_sort(synth.fuzz_split_reads(22, 300, refs3, lens3) + synth.planted_reads(23, 400, ref, refs3, lens3, n_sites=30, types=("DEL", "INS", "INV")))
    p2 = str(tmp_path / "split.bam")
    records.write_bam(p2, refs3, lens3, rr)
    files.append((p2, 97))
    short, long_rec, cig = H.long_cigar_records()
    p3 = str(tmp_path / "cg.bam")
    records.write_bam(p3, ["chr1"], [2000000], [short, long_rec, short])
    files.append((p3, 5))
    truth = {p1: recs[:3000], p2: rr, p3: [short, long_rec, short]}
    for path, per_batch in files:
        host = NativeBam(path, threads=2)
        want, want_names = _read_all_batches(host, per_batch)
        host.close()
        dev = NativeBam(path, threads=2)
        dev.set_device_decode(0)
        for rep in range(2):                                               # a second pass after rewind: same records, same ids
            got, got_names = _read_all_batches(dev, per_batch)
            assert got_names == want_names, path
            a, b = _concat_batches(got), _concat_batches(want)
            assert len(a) == len(b) and len(a) > 2, path
            # not only "equal to the host reader": the records the file was written from (two product paths that agree on a wrong answer cannot pass)
            assert got_names == [r.query_name for r in truth[path]], path
            H.assert_rows_are_the_written_records(a, truth[path], path)
            for k, (x, y) in enumerate(zip(a, b)):
                assert x == y, (path, k, [i for i, (u, v) in enumerate(zip(x, y)) if u != v])
            dev.rewind()
        dev.close()


def _expected_query_slots(A):
    """emission slots of a query-name batch recomputed from its flags, read ids and segment offsets (src/svim/SVIM_COLLECT.py:114-121: per analysed read
    primary, good supplementaries.., segments)"""
    n = len(A["flag"])
    order, seg_order = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
    slot, i = 0, 0
    while i < n:
        j = i
        while j < n and A["read_id"][j] == A["read_id"][i]:
            j += 1
        live = [k for k in range(i, j) if not (int(A["flag"][k]) & 0x8000)]
        if live:
            prim = [k for k in live if not (int(A["flag"][k]) & 2048)]
            assert len(prim) == 1
            sups = [k for k in live if int(A["flag"][k]) & 2048]
            order[prim[0]] = slot
            for q, k in enumerate(sups):
                order[k] = slot + 1 + q
            seg_order[prim[0]] = slot + 1 + len(sups)
            assert int(A["seg_off"][prim[0] + 1]) - int(A["seg_off"][prim[0]]) == len(sups)
            slot += len(sups) + 2
        i = j
    return order, seg_order


@pytest.mark.parametrize("chunk_blocks", [None, "1", "2"])
def test_device_bam_decode_queryname_mode_equals_host_reader(tmp_path, monkeypatch, chunk_blocks):
    """Query-name-sorted input on the device-resident reader (src/svim/SVIM_COLLECT.py:8-41 bam_iterator, :96-129): read groups, the "exactly one good primary"
    rule, SVX_FLAG_SKIP, the segment table made of the good supplementary RECORDS and the per-read emission slots come out as the host reader makes them
    (csrc/bamio.cpp) - on a fuzz file with every kind of group (no primary, two primaries, unmapped or low-mapq primary, secondary records, low-mapq
    supplementaries), whole file in one chunk and in chunks of one or two BGZF blocks (groups and records straddle chunk boundaries), small batches
    (groups are never split), two passes."""
    from svim_amd.bamio import NativeBam
    if chunk_blocks:
        monkeypatch.setenv("SVX_BAM_DEV_CHUNK_BLOCKS", chunk_blocks)
    refs, lens = ["chr1", "chr2", "chr10"], [180000, 60000, 60000]
    ref = synth.make_reference(21, list(zip(refs, lens)))
    recs = synth.fuzz_split_reads(24, 420, refs, lens) + synth.planted_reads(25, 260, ref, refs, lens, n_sites=20, types=("DEL", "INS", "INV"))
    rng = random.Random(8)
    groups = {}
    for a in recs:
        groups.setdefault(a.query_name, []).append(a)
    names = list(groups)
    rng.shuffle(names)
    out = []
    for k, nm in enumerate(names):
        g = groups[nm]
        rng.shuffle(g)                                                     # the primary is not always first
        if k % 17 == 3 and len(g) > 1:                                     # a second primary: the read is not analysed (:108)
            g[1].flag &= ~(256 | 2048)
        if k % 19 == 5:
            g[0].flag |= 4                                                 # an unmapped record
        if k % 23 == 7:
            for a in g:
                a.mapping_quality = 3                                      # below min_mapq
        out += g
    path = str(tmp_path / "q.bam")
    records.write_bam(path, refs, lens, out, sort_order="queryname")
    host = NativeBam(path, threads=2)
    want, want_names = [], []
    while True:
        b, n = host.read_batch(61, 20, "queryname")
        if n == 0:
            break
        A = host.batch_arrays(b)
        nm = host.read_names()
        want_names += [nm[int(i)] for i in A["read_id"]]
        want.append(A)
    host.close()
    n_live = sum(int(((A["flag"] & 0x8000) == 0).sum()) for A in want)
    n_rows = sum(int(A["seg_off"][-1]) for A in want)
    assert n_live > 300 and n_rows > 100 and sum(len(A["flag"]) for A in want) > n_live + 50
    dev = NativeBam(path, threads=2)
    dev.set_device_decode(0)
    for rep in range(2):
        got, got_names = [], []
        held = None
        while True:
            b, n = dev.read_batch(61, 20, "queryname")
            if n == 0:
                break
            A = dev.batch_arrays(b)
            if held is not None:
                # include/svx.h: the per-batch arrays of query-name mode (flag, order, seg_order, the seg_* table) alternate between two sets - the batch handed
                # out BEFORE this one is still intact (a consumer may stay one batch behind the reader, as harness.BamPipeline does)
                again = dev.batch_arrays(held[0])
                assert all(np.array_equal(again[k], held[1][k]) for k in held[1]), (chunk_blocks, rep, [k for k in held[1] if not np.array_equal(again[k], held[1][k])])
            held = (b, A) if chunk_blocks is None else None                # (small chunks: a batch may be the last of its chunk - its slot rotates on)
            nm = dev.read_names()
            got_names += [nm[int(i)] for i in A["read_id"]]
            got.append(A)
            exp_o, exp_s = _expected_query_slots(A)
            assert np.array_equal(A["order"], exp_o) and np.array_equal(A["seg_order"], exp_s), (chunk_blocks, rep)
        assert got_names == want_names
        cuts = np.cumsum([len(x["flag"]) for x in got])[:-1]
        assert all(got_names[int(c) - 1] != got_names[int(c)] for c in cuts)      # a read's group is whole inside its batch
        a, b = H.concat_batch_rows(got), H.concat_batch_rows(want)
        assert len(a) == len(b)
        H.assert_rows_are_the_written_records(a, out, "query-name file")      # (not only the host reader's answer: the records the file was written from)
        assert got_names == [r.query_name for r in out]
        bad = [k for k, (x, y) in enumerate(zip(a, b)) if x != y]
        assert not bad, (chunk_blocks, rep, bad[:5], a[bad[0]][:5], b[bad[0]][:5])
        if chunk_blocks is None:                                           # one chunk: the batches themselves are the host reader's
            assert [len(A["flag"]) for A in got] == [len(A["flag"]) for A in want]
            for A, B in zip(got, want):
                assert np.array_equal(A["order"], B["order"]) and np.array_equal(A["seg_order"], B["seg_order"]) and np.array_equal(A["seg_off"], B["seg_off"])
        dev.rewind()
    dev.close()


@pytest.mark.parametrize("name", ["fuzzA", "fuzzB", "fuzzC", "layoutD"])
def test_queryname_golden_cases_through_a_bam_file_on_the_device_reader(eng, tmp_path, name):
    """the reference's own outputs for query-name-sorted input (tests/golden/g2_collect.json.gz, generated by running src/svim/SVIM_COLLECT.py:96-129) through a BAM file,
    the device-resident reader and svx_collect, in small batches"""
    from svim_amd import SVIM_COLLECT
    g = H.load("g2_collect.json.gz")
    cases = [c for c in g["cases"] if c["name"] == name and c["mode"] == "queryname"]
    text = [c for c in cases if c.get("sam")][0]["sam"]
    bam = records.AlignmentFile(text=text)
    recs = list(bam.fetch(until_eof=True))
    path = str(tmp_path / "q.bam")
    records.write_bam(path, bam.references, bam.lengths, recs, sort_order="queryname")
    for case in cases:
        o = H.options(case["options"])
        sigs, bnds = SVIM_COLLECT._run_native(path, o, "queryname", batch_records=23)
        assert [H.sig_row(s) for s in sigs] == case["signatures"]
        assert [H.sig_row(s) for s in bnds] == case["bnds"]


def test_device_bam_decode_regions_and_pipeline(eng, tmp_path, monkeypatch):
    """contig-range reading (svx_bam_seek + reference id limit) in device mode == host mode, out of file order; and BamPipeline on the device reader
    accumulates the same signature list as on the host reader."""
    from svim_amd import harness
    from svim_amd.bamio import NativeBam
    refs, lens = ["chr1", "chr2", "chr10", "chr3"], [100000, 80000, 80000, 60000]
    ref = synth.make_reference(61, list(zip(refs, lens)))
    recs = synth.coordinate_sort(synth.fuzz_split_reads(62, 260, refs, lens, max_sv_size=20000) + synth.planted_reads(63, 300, ref, refs, lens, n_sites=25, types=("DEL", "INS", "INV")))
    path = str(tmp_path / "m.bam")
    records.write_bam(path, refs, lens, recs)
    bai = records.read_bai(path + ".bai")
    regions = [(bai[t][0], t) for t in (2, 0, 3, 1) if bai[t] is not None]
    host = NativeBam(path, threads=2)
    want, want_names = _read_all_batches(host, 41, regions)
    host.close()
    monkeypatch.setenv("SVX_BAM_DEV_CHUNK_BLOCKS", "2")
    dev = NativeBam(path, threads=2)
    dev.set_device_decode(0)
    got, got_names = _read_all_batches(dev, 41, regions)
    dev.close()
    assert got_names == want_names
    assert _concat_batches(got) == _concat_batches(want)
    # not only "equal to the host reader": the regions, in the order asked for, are the records the file was WRITTEN from (a region = every record of its contig)
    truth = [r for t in (2, 0, 3, 1) if bai[t] is not None for r in recs if r.reference_id == t]
    assert got_names == [r.query_name for r in truth]
    H.assert_rows_are_the_written_records(_concat_batches(got), truth, "regions")
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 20000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5, "partition_max_distance": 1000,
                   "position_distance_normalizer": 900, "edit_distance_normalizer": 1.0, "cluster_max_distance": 0.5, "all_bnds": False})
    tabs = []
    for device_decode in (True, False):
        pipe = harness.BamPipeline(path, o, eng, threads=2, batch_records=53, device_decode=device_decode)
        assert pipe.device_decode is device_decode
        n = pipe.run()
        assert n == len(recs)
        names = pipe.bam.read_names()
        sig = eng.fetch_signatures(0)
        pipe.close()
        tabs.append([(int(sig.type[i]), int(sig.contig[i]), int(sig.start[i]), int(sig.end[i]), int(sig.contig2[i]), int(sig.pos2[i]), names[int(sig.read_id[i])], sig.sequence(i))
                     for i in range(sig.n)])
    assert tabs[0] == tabs[1] and len(tabs[0]) > 100


def test_device_batches_stay_valid_across_seek_and_rewind(tmp_path, monkeypatch):
    """include/svx.h: the arrays of a batch stay valid until the third next chunk is loaded - also when a seek or a rewind comes in between (the reader
    thread of BamPipeline runs one batch ahead of the GPU: a seek used to load the new region into the slot the last batch still lived in).  The batch
    handed out last is only copied AFTER the next region's first batch has been read (which loads a chunk and prefetches another); regions of a few blocks,
    chunks of two blocks."""
    from svim_amd.bamio import NativeBam
    refs, lens = ["chr1", "chr2", "chr10", "chr3"], [100000, 80000, 80000, 60000]
    ref = synth.make_reference(71, list(zip(refs, lens)))
    recs = synth.coordinate_sort(synth.fuzz_split_reads(72, 200, refs, lens, max_sv_size=20000) + synth.planted_reads(73, 260, ref, refs, lens, n_sites=20, types=("DEL", "INS")))
    path = str(tmp_path / "m.bam")
    records.write_bam(path, refs, lens, recs)
    bai = records.read_bai(path + ".bai")
    order = [t for t in (2, 0, 3, 1, 2, 0) if bai[t] is not None]
    host = NativeBam(path, threads=2)
    want = []
    for t in order:
        host.seek(bai[t][0], t)
        b, n = host.read_batch(37, 20, "coordinate")
        assert n > 0
        want.append(_concat_batches([host.batch_arrays(b)]))
    host.rewind()
    b, n = host.read_batch(37, 20, "coordinate")
    want_first = _concat_batches([host.batch_arrays(b)])
    host.close()
    # the expectation itself is held to the records the file was written from: the first records of every contig asked for, and of the file after the rewind
    for k, t in enumerate(order):
        H.assert_rows_are_the_written_records(want[k], [r for r in recs if r.reference_id == t][:len(want[k])], "region of contig %d" % t)
    H.assert_rows_are_the_written_records(want_first, recs[:len(want_first)], "after rewind")
    monkeypatch.setenv("SVX_BAM_DEV_CHUNK_BLOCKS", "2")
    dev = NativeBam(path, threads=2)
    dev.set_device_decode(0)
    held = None
    for k, t in enumerate(order):
        dev.seek(bai[t][0], t)
        b, n = dev.read_batch(37, 20, "coordinate")
        assert n > 0
        if held is not None:
            _same_leading_records(dev.batch_arrays(held), want[k - 1])              # the previous region's batch, copied only now
        held = b
    dev.rewind()
    b, n = dev.read_batch(37, 20, "coordinate")
    _same_leading_records(dev.batch_arrays(held), want[-1])
    _same_leading_records(dev.batch_arrays(b), want_first)
    dev.close()


def _same_leading_records(A, want_rows):
    """a device batch ends where its chunk ends, so it may hold fewer records than the host reader's batch of the same request: the records it holds are
    the leading ones"""
    got = _concat_batches([A])
    assert 0 < len(got) <= len(want_rows) and got == want_rows[:len(got)]


def _bgzf_offsets(raw):
    """(offset, total size) of every BGZF block of a file image (SAM spec 4.1: BSIZE at byte 16 of a block with the standard 6-byte extra field)"""
    out, at = [], 0
    while at + 18 <= len(raw):
        size = int.from_bytes(raw[at + 16:at + 18], "little") + 1
        out.append((at, size))
        at += size
    return out


def test_device_bam_decode_edge_cases_and_damaged_files(tmp_path, monkeypatch):
    """The device-resident reader on inputs at the edges: a file with a header and no record, unmapped records behind the mapped ones (both readers hand them out, COLLECT
    skips them by their flag), a file cut in the middle of a BGZF block, a block whose DEFLATE stream is damaged, a block whose
    ISIZE lies, a block that inflates to the wrong bytes (only its CRC32 tells) - errors are raised (SvxError), nothing hangs, nothing is handed out silently."""
    from svim_amd import _lib
    from svim_amd.bamio import NativeBam
    g, refs, recs = H.c1_case()
    # (a) header only
    p0 = str(tmp_path / "empty.bam")
    records.write_bam(p0, ["chr1"], [2000000], [])
    for device in (False, True):
        nb = NativeBam(p0, threads=2)
        if device:
            nb.set_device_decode(0)
        b, n = nb.read_batch(100, 20, "coordinate")
        assert n == 0
        nb.close()
    # (b) unmapped records at the end
    import copy
    tail = []
    for r in recs[:5]:
        u = copy.copy(r)
        u.reference_id, u.reference_start, u.flag, u.cigartuples, u.mapping_quality = -1, -1, 4, [], 0
        u.query_name = "unmapped_" + r.query_name
        tail.append(u)
    p1 = str(tmp_path / "tail.bam")
    records.write_bam(p1, ["chr1"], [2000000], recs[:400] + tail)
    host = NativeBam(p1, threads=2)
    want, want_names = _read_all_batches(host, 150)
    host.close()
    dev = NativeBam(p1, threads=2)
    dev.set_device_decode(0)
    got, got_names = _read_all_batches(dev, 150)
    dev.close()
    assert got_names == want_names
    assert _concat_batches(got) == _concat_batches(want)
    assert len(want_names) == 405 and want_names[-1].startswith("unmapped_")       # handed out like any record (COLLECT skips them by flag)
    # (c) (d) (e) damaged files
    p2 = str(tmp_path / "good.bam")
    records.write_bam(p2, ["chr1"], [2000000], recs[:1500])
    raw = open(p2, "rb").read()
    blocks = _bgzf_offsets(raw)
    assert len(blocks) > 6
    mid_at, mid_size = blocks[len(blocks) // 2]
    cut = raw[:mid_at + mid_size // 2]
    bad_type = bytearray(raw)
    bad_type[mid_at + 18] |= 0x06                                        # BTYPE = 3 (reserved) in the block's first DEFLATE header
    bad_isize = bytearray(raw)
    bad_isize[mid_at + mid_size - 4:mid_at + mid_size] = (int.from_bytes(raw[mid_at + mid_size - 4:mid_at + mid_size], "little") - 7).to_bytes(4, "little")
    # a block that inflates without complaint to the right length but to the wrong bytes: one quality byte changed, the trailer's CRC32 kept
    import zlib
    pay = zlib.decompress(raw[mid_at + 18:mid_at + mid_size - 8], -15)
    run = pay.find(b"\xff" * 40)
    assert run > 0
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(pay[:run + 20] + b"\xfe" + pay[run + 21:]) + co.flush()
    wrong = bytearray(raw[mid_at:mid_at + 18]) + body + raw[mid_at + mid_size - 8:mid_at + mid_size]
    wrong[16:18] = (len(wrong) - 1).to_bytes(2, "little")
    bad_crc = raw[:mid_at] + bytes(wrong) + raw[mid_at + mid_size:]
    for name, image in (("cut", cut), ("btype", bytes(bad_type)), ("isize", bytes(bad_isize)), ("crc", bad_crc)):
        path = str(tmp_path / (name + ".bam"))
        with open(path, "wb") as fh:
            fh.write(image)
        for device in (True, False):
            if name == "crc" and not device:
                continue                                                   # (the host reader verifies CRCs only with SVX_BAM_VERIFY_CRC=1: tests/test_host_cpu.py)
            nb = None
            with pytest.raises(_lib.SvxError):                             # (a cut file is refused when it is opened already)
                nb = NativeBam(path, threads=2)
                if device:
                    nb.set_device_decode(0)
                while True:
                    b, n = nb.read_batch(200, 20, "coordinate")
                    if n == 0:
                        break
            if nb is not None:
                nb.close()


def test_own_radix_sort_and_scan(eng):
    """csrc/prims.hip, csrc/scan.hpp (hand-written; rocPRIM in rounds 1-2): stable LSD radix sort of (u64 key, u32 value) pairs over a bit range and the
    exclusive scan, against std::stable_sort / a serial sum - sizes around every path switch (one workgroup <= 16384 pairs, tiles of 2048; scan: one
    launch <= 8192), the bit ranges the path uses (0..32, 0..40, 0..64) and ragged ones; seeds with bit 8 set make every key narrow (passes over a
    constant digit are skipped in the one-workgroup form)."""
    for n in (0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 8191, 8192, 8193, 10000, 16383, 16384, 16385, 32767, 32768, 32769, 100003, 1 << 20, 3333333):
        for b0, b1 in ((0, 64), (0, 32), (0, 40)):
            eng.selftest_prims(n, b0, b1, seed=(n + b1) & ~0x100)
            eng.selftest_prims(n, b0, b1, seed=(n + b1) | 0x100)
    for b0, b1 in ((3, 17), (8, 9), (60, 64), (31, 33), (0, 1), (5, 64)):
        for n in (100, 5000, 8192, 70001):
            eng.selftest_prims(n, b0, b1, seed=7 * n + b0)
