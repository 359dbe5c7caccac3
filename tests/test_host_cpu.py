"""CPU tests: host logic (records, batcher, object layer, multi-GPU merge) and the C ABI surface.
No compute call is made on the product library here (it has no CPU path)."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers as H
from svim_amd import _abi, _lib, batch, convert, records, synth
from svim_amd.SVSignature import SignatureDeletion, SignatureInsertion


def test_abi_library_exports_every_declared_symbol():
    _lib.build()
    assert os.path.exists(_lib._LIB_PATH)
    L = ctypes.CDLL(_lib._LIB_PATH)
    header = open(os.path.join(os.path.dirname(_lib._HERE), "include", "svx.h")).read()
    declared = sorted(set(re.findall(r"\b(svx_[a-z0-9_]+)\s*\(", header)))
    assert set(declared) == set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.SvxError):
        _lib.Engine(0)
    import svim_amd
    with pytest.raises(_lib.SvxError):
        svim_amd.analyze_cigar_indel([(0, 10)], 5)


def test_signature_accessors_like_reference_tests():
    # src/tests/test_Signature.py:7-29
    d = SignatureDeletion("chr1", 100, 300, "cigar", "read1")
    assert d.get_source() == ("chr1", 100, 300)
    assert d.get_key() == ("DEL", "chr1", 300)
    d2 = SignatureDeletion("chr1", 450, 500, "cigar", "read2")
    d3 = SignatureDeletion("chr1", 150, 200, "cigar", "read3")
    d4 = SignatureDeletion("chr2", 350, 400, "cigar", "read3")
    ins = SignatureInsertion("chr1", 150, 200, "cigar", "read2", "ACGT")
    assert d.downstream_distance_to(d2) == 150
    assert d.downstream_distance_to(d3) == 0
    assert d.downstream_distance_to(d4) == float("inf")
    assert d.downstream_distance_to(ins) == float("inf")
    assert d.as_string() == "chr1\t100\t300\tDEL;cigar\tread1"
    assert d.as_string(":") == "chr1:100:300:DEL;cigar:read1"


def test_is_similar_like_reference_tests():
    # src/tests/test_inter.py:7-11
    from svim_amd.SVIM_inter import is_similar
    assert not is_similar("chrI", 0, 100, "chrII", 0, 100)
    assert is_similar("chrI", 0, 100, "chrI", 0, 100)
    assert is_similar("chrI", 0, 100, "chrI", 10, 90)
    assert not is_similar("chrI", 100, 105, "chrI", 21, 100)


def test_object_strings_match_reference_golden():
    """as_string / get_key / partition gaps of our classes on every golden signature row."""
    g = H.load("g5_cluster.json.gz")
    rows = g["cases"][-1]["signatures"]
    for r in rows[:3000]:
        s = H.row_sig(r)
        assert H.sig_row(s) == r
        assert isinstance(s.as_string(), str) and s.get_key()[0] == r[0]


def test_form_partitions_host_helper_matches_golden():
    from svim_amd.SVIM_clustering import form_partitions
    g4 = H.load("g4_partitions.json.gz")
    g5 = H.load("g5_cluster.json.gz")
    cases = {c["name"]: c for c in g5["cases"]}
    p = g4["partitions"][0]
    case = cases[p["case"]]
    sigs = [H.row_sig(r) for r in case["signatures"]]
    idx = {id(s): i for i, s in enumerate(sigs)}
    sub = [s for s in sigs if s.type == p["type"]]
    got = [[idx[id(s)] for s in q] for q in form_partitions(sub, case["options"]["partition_max_distance"])]
    assert got == p["partitions"]


def test_satag_reconstruction_like_reference_test():
    # src/tests/test_satag.py:16-34 replayed on our record model
    from svim_amd.SVIM_COLLECT import retrieve_other_alignments
    bam = records.AlignmentFile(os.path.join(H.GOLDEN, "chimeric_read.sam"))
    alns = list(bam.fetch(until_eof=True))
    assert len(alns) == 4
    sup = retrieve_other_alignments(alns[0], bam)
    assert len(sup) == 3
    for a, b in zip(sup, alns[1:]):
        assert a.cigarstring == b.cigarstring and a.reference_id == b.reference_id
        assert a.reference_start == b.reference_start and a.reference_end == b.reference_end
        assert a.flag == b.flag and a.mapping_quality == b.mapping_quality
        assert a.query_name == b.query_name
        assert a.query_alignment_start == b.query_alignment_start and a.query_alignment_end == b.query_alignment_end


def test_bam_roundtrip_and_iterator(tmp_path):
    # src/tests/test_Collect.py:252-268 shape: 10 primary-only + 10 primary+supplementary reads, grouped by name
    from svim_amd.SVIM_COLLECT import bam_iterator
    refs, lens = ["chr1", "chr2"], [300000, 200000]
    recs = synth.fuzz_split_reads(5, 30, refs, lens)
    path = str(tmp_path / "t.bam")
    records.write_bam(path, refs, lens, recs, sort_order="queryname")
    bam = records.AlignmentFile(path)
    back = list(bam.fetch(until_eof=True))
    assert len(back) == len(recs)
    for a, b in zip(recs, back):
        assert (a.query_name, a.flag, a.reference_id, a.reference_start, a.mapping_quality) == \
               (b.query_name, b.flag, b.reference_id, b.reference_start, b.mapping_quality)
        assert a.cigartuples == b.cigartuples and a.query_sequence == b.query_sequence
        assert a.get_tags() == b.get_tags()
    assert bam.header["HD"]["SO"] == "queryname"
    groups = list(bam_iterator(bam))
    assert len(groups) == 30 and all(len(set(x.query_name for x in p + s + q)) == 1 for p, s, q in groups)


def test_batch_builder_layout():
    refs, lens = ["chr1", "chr2", "chr10"], [180000, 60000, 60000]
    recs = synth.coordinate_sort(synth.fuzz_split_reads(9, 40, refs, lens))
    bam = records.AlignmentFile(text=synth.sam_text(refs, lens, recs))
    hb = batch.build_batch(bam, H.options({"min_mapq": 20}), mode="coordinate")
    A = hb.arrays
    assert hb.n_rec == len(recs)
    assert A["cigar_off"][-1] == sum(len(a.cigartuples or []) for a in recs)
    assert list(A["contig_rank"]) == [0, 2, 1]          # 'chr1' < 'chr10' < 'chr2'
    for i, a in enumerate(recs):
        lo, hi = int(A["cigar_off"][i]), int(A["cigar_off"][i + 1])
        assert [(int(c) & 15, int(c) >> 4) for c in A["cigar"][lo:hi]] == (a.cigartuples or [])
        if a.query_sequence:
            packed = A["seq"][int(A["seq_off"][i]):int(A["seq_off"][i + 1])]
            s = "".join(_abi.NIBBLE[b >> 4] + _abi.NIBBLE[b & 15] for b in packed)[:len(a.query_sequence)]
            assert s == a.query_sequence
    # segment rows only for usable primaries carrying an SA tag
    for i, a in enumerate(recs):
        nseg = int(A["seg_off"][i + 1]) - int(A["seg_off"][i])
        usable = not (a.flag & (4 | 256 | 2048)) and a.mapping_quality >= 20 and a.has_tag("SA")
        assert (nseg > 0) == usable


def test_table_object_roundtrip():
    g = H.load("g5_cluster.json.gz")
    rows = g["cases"][0]["signatures"]
    sigs = [H.row_sig(r) for r in rows]
    tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(H.REFS))
    back = convert.objects_from_sigtable(tab, contigs.names, reads.names)
    assert [H.sig_row(s) for s in back] == rows
    with pytest.raises(TypeError):
        convert.sigtable_from_objects([SignatureDeletion("chr1", 10.5, 20.5, "cigar", "r")])
    with pytest.raises(ValueError):
        convert.sigtable_from_objects([SignatureInsertion("chr1", 10, 20, "cigar", "r", "ACGU")])


@pytest.mark.parametrize("mode", ["coordinate", "queryname"])
def test_native_bam_reader_matches_python_batcher(tmp_path, mode):
    """The C++ BAM front-end (BGZF inflate + record decode + SA / grouping rules) builds the same record batch as the
    Python reader + batcher, field by field; also when the file is consumed in several batches."""
    from svim_amd.bamio import NativeBam
    refs, lens = ["chr1", "chr2", "chr10"], [180000, 60000, 60000]
    recs = synth.fuzz_split_reads(21, 120, refs, lens)
    recs += synth.planted_reads(22, 60, synth.make_reference(1, list(zip(refs, lens))), refs, lens, n_sites=10)
    if mode == "coordinate":
        recs = synth.coordinate_sort(recs)
    else:
        # keep each read's records together
        order = {}
        for a in recs:
            order.setdefault(a.query_name, len(order))
        recs = sorted(recs, key=lambda a: order[a.query_name])
    path = str(tmp_path / "t.bam")
    records.write_bam(path, refs, lens, recs, sort_order=mode)
    o = H.options({"min_mapq": 20})
    hb = batch.build_batch(records.AlignmentFile(path), o, mode=mode)
    nb = NativeBam(path, threads=3)
    assert nb.references == refs and nb.lengths == lens and nb.sort_order == mode
    b, n = nb.read_batch(1 << 30, 20, mode)
    assert n == hb.n_rec and b.n_seg == hb.n_seg
    A = nb.batch_arrays(b)
    for k in _abi.BATCH_DTYPES:
        exp = hb.arrays[k]
        got = A[k]
        assert np.array_equal(got, exp[:got.size]), k
    assert nb.read_names() == hb.read_names
    b2, n2 = nb.read_batch(10, 20, mode)
    assert n2 == 0
    nb.close()
    # several batches: concatenation of the per-batch records is the file
    nb = NativeBam(path, threads=2)
    tot, flags = 0, []
    while True:
        b, n = nb.read_batch(37, 20, mode)
        if n == 0:
            break
        A = nb.batch_arrays(b)
        flags.append(A["flag"] & 0x0fff)
        tot += n
        if mode == "queryname":        # a read is never split across batches
            ids = A["read_id"]
            assert len(set(ids.tolist())) == len([1 for i in range(len(ids)) if i == 0 or ids[i] != ids[i - 1]])
    assert tot == hb.n_rec
    assert np.array_equal(np.concatenate(flags), hb.arrays["flag"] & 0x0fff)
    nb.close()


@pytest.mark.parametrize("env", [{"SVX_BAM_CHUNK_BLOCKS": "1"}, {"SVX_BAM_CHUNK_BLOCKS": "2", "SVX_BAM_WIN_HEAD": "16"},
                                 {"SVX_BAM_CHUNK_BLOCKS": "3", "SVX_BAM_WIN_HEAD": "0"}])
def test_native_bam_reader_chunk_switches(tmp_path, monkeypatch, env):
    """The reader inflates the file in chunks on a background thread and carries the partial record at the end of a chunk into the next
    window (through the headroom, or by appending when it does not fit).  Tiny chunks / tiny headroom exercise both paths on
    every switch; the result must not depend on them."""
    from svim_amd.bamio import NativeBam
    refs, lens = ["chr1", "chr2"], [400000, 60000]
    recs = synth.coordinate_sort(synth.fuzz_split_reads(5, 400, refs, lens, read_len=(3000, 30000)))
    path = str(tmp_path / "t.bam")
    records.write_bam(path, refs, lens, recs, sort_order="coordinate")
    assert os.path.getsize(path) > 400000                     # dozens of BGZF blocks

    def read_all(batch_size):
        nb = NativeBam(path, threads=4)
        out = {k: [] for k in ("flag", "pos", "lseq", "cigar", "seq", "read_id")}
        n_tot = 0
        while True:
            b, n = nb.read_batch(batch_size, 20, "coordinate")
            if n == 0:
                break
            A = nb.batch_arrays(b)
            for k in out:
                out[k].append(A[k].copy())
            n_tot += n
        names = nb.read_names()
        nb.close()
        return n_tot, {k: np.concatenate(v) for k, v in out.items()}, names

    ref_n, ref_arrays, ref_names = read_all(1 << 30)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for batch_size in (1 << 30, 50):
        n, arrays, names = read_all(batch_size)
        assert n == ref_n == len(recs) and names == ref_names
        for k in arrays:
            assert np.array_equal(arrays[k], ref_arrays[k]), (k, batch_size)


def _stress31_objects(oracle):
    """g5's 'stress31' case through the CPU stand-in engine (the oracle) -> our signature objects + the lazy cluster 6-tuple."""
    g5 = H.load("g5_cluster.json.gz")
    case = [c for c in g5["cases"] if c["name"] == "stress31"][0]
    o = H.options(case["options"])
    sigs = [H.row_sig(r) for r in case["signatures"]]
    tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(g5["references"]))
    off, codes = convert.genome_arrays(o.genome, contigs.names)
    oracle.set_genome(off, codes)
    ct = oracle.cluster(_abi.Params.from_options(o), batch.contig_ranks(contigs.names), table=tab)
    return case, tab, contigs, reads, sigs, ct


def test_writers_match_reference_writer_text(oracle, tmp_path):
    """write_signature_clusters_bed / _vcf (src/svim/SVIM_CLUSTER.py:29-106): the files the reference's writers produced for the
    reference's clusters (tests/golden/g_writers.json.gz) vs our writers on our objects; FP columns within 1e-9."""
    from svim_amd import SVIM_CLUSTER
    gold = H.load("g_writers.json.gz")
    case, tab, contigs, reads, sigs, ct = _stress31_objects(oracle)
    clusters = convert.cluster_objects(ct, sigs, contigs.names)
    SVIM_CLUSTER.write_signature_clusters_bed(str(tmp_path), clusters)
    SVIM_CLUSTER.write_signature_clusters_vcf(str(tmp_path), clusters, gold["version"])
    import os
    for name, exp in gold["files"].items():
        with open(os.path.join(str(tmp_path), name)) as fh:
            got = fh.read()
        assert exp.strip() != "" or name.endswith(".bed")
        assert H.text_close(got, exp) is None, (name, H.text_close(got, exp))
    assert sum(len(t) for t in gold["files"].values()) > 100000


def test_lazy_lists_build_objects_only_when_read(oracle):
    from svim_amd.lazy import ClusterList, SignatureList
    case, tab, contigs, reads, sigs, ct = _stress31_objects(oracle)
    lazy = SignatureList(tab, contigs.names, reads.names)
    assert len(lazy) == len(sigs) and lazy._objs is None
    assert lazy.count_by_type()["INS"] == sum(1 for s in sigs if s.type == "INS")
    assert H.sig_row(lazy[5]) == H.sig_row(sigs[5]) and H.sig_row(lazy[-1]) == H.sig_row(sigs[-1])
    assert lazy._objs is None and len(lazy._one) == 2                      # single reads do not build the whole list
    clusters = convert.cluster_objects(ct, lazy, contigs.names)
    assert all(isinstance(c, ClusterList) for c in clusters) and lazy._objs is None
    assert [len(c) for c in clusters] == [len(x) for x in case["clusters"]]
    first = clusters[1][0]                                                 # building a cluster object leaves its members alone
    assert type(first._members) is tuple and lazy._objs is None
    m = first.members
    assert [H.sig_row(s) for s in m] == [case["signatures"][j] for j in case["clusters"][1][0][7]]
    assert first.members is m                                              # resolved once
    with pytest.raises(IndexError):
        lazy[len(sigs)]
    # iteration builds everything in one vectorised pass, in list order; reference-style filters work
    assert [H.sig_row(s) for s in lazy] == case["signatures"] and lazy._objs is not None
    assert len([ev for ev in lazy if ev.type == "DEL"]) == lazy.count_by_type()["DEL"]
    assert lazy[2:4] == lazy.materialise()[2:4] and (lazy + [1])[-1] == 1
    # the whole reference tuple, rows compared with the golden
    rows = []
    for k, lst in enumerate(clusters):
        idx = {id(s): i for i, s in enumerate(lazy)}
        rows.append([[idx[id(x)] for x in c.members] for c in lst])
    exp = [[r[7] if k < 3 else r[10] for r in case["clusters"][k]] for k in range(6)]
    assert rows == exp


def test_native_bam_reader_sparse_seq_and_double_buffer(tmp_path, oracle):
    """svx_bam_set_seq_filter: the reader keeps only the SEQ ranges COLLECT reads (insertions >= min_sv_size, whole split-read
    primaries).  The oracle's COLLECT on the sparse batches must equal its COLLECT on the dense ones, sequences included, with a
    fraction of the bases; and a batch stays intact while the next one is read (two array sets alternate)."""
    from svim_amd.bamio import NativeBam
    refs, lens = ["chr1", "chr2", "chr10"], [180000, 60000, 60000]
    ref = synth.make_reference(1, list(zip(refs, lens)))
    recs = synth.coordinate_sort(synth.fuzz_split_reads(31, 150, refs, lens) +
                                 synth.planted_reads(32, 220, ref, refs, lens, n_sites=30, types=("DEL", "INS", "INV")))
    path = str(tmp_path / "s.bam")
    records.write_bam(path, refs, lens, recs)
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5,
                   "partition_max_distance": 1000, "position_distance_normalizer": 900, "edit_distance_normalizer": 1.0,
                   "cluster_max_distance": 0.5, "all_bnds": True})
    p = _abi.Params.from_options(o)

    def collect_all(filter_len, per_batch):
        nb = NativeBam(path, threads=3)
        if filter_len:
            nb.set_seq_filter(filter_len)
        out, kept, prev = [], 0, None
        while True:
            b, n = nb.read_batch(per_batch, 20, "coordinate")
            if n == 0:
                break
            if prev is not None:                    # the previous batch's arrays are still what they were
                pb, snap = prev
                assert np.array_equal(nb.batch_arrays(pb)["cigar"], snap)
            A = nb.batch_arrays(b)
            kept += A["seq"].size
            assert (b.seq_rng_off is not None) == bool(filter_len)
            s, t = oracle.collect(b, p)
            out.append((s, t))
            prev = (b, A["cigar"].copy())
        nb.close()
        return out, kept
    dense, dense_bytes = collect_all(0, 90)
    sparse, sparse_bytes = collect_all(40, 90)
    assert len(dense) == len(sparse) >= 4
    n_ins = 0
    for (s0, t0), (s1, t1) in zip(dense, sparse):
        assert s1.first_difference(s0) is None and t1.first_difference(t0) is None
        n_ins += int((s0.type[:s0.n] == 1).sum())
    assert n_ins > 50 and sparse_bytes < 0.7 * dense_bytes
    # a filter larger than min_sv_size loses bases COLLECT needs: loud failure, not silence
    nb = NativeBam(path, threads=2)
    nb.set_seq_filter(400)
    b, n = nb.read_batch(1 << 30, 20, "coordinate")
    with pytest.raises(AssertionError):
        oracle.collect(b, p)
    nb.close()


def test_long_cigar_cg_tag_through_both_readers(tmp_path, oracle):
    """A CIGAR beyond 65535 operations does not fit the BAM record's 16-bit count: it lives in the CG:B,I tag behind a <l_seq>S<ref_len>N
    placeholder (SAM spec).  Both the Python reader (records.py) and the native one (bamio.cpp, dense and
    sparse SEQ) must hand COLLECT the real CIGAR: same batch arrays, and the planted insertion / deletion come out with their bases."""
    from svim_amd.bamio import NativeBam
    short, a, cig = H.long_cigar_records()
    path = str(tmp_path / "cg.bam")
    records.write_bam(path, ["chr1"], [400000], [short, a])
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5,
                   "partition_max_distance": 1000, "position_distance_normalizer": 900, "edit_distance_normalizer": 1.0,
                   "cluster_max_distance": 0.5, "all_bnds": False})
    p = _abi.Params.from_options(o)
    back = list(records.AlignmentFile(path).fetch(until_eof=True))
    assert back[1].cigartuples == cig and back[1].query_sequence == a.query_sequence
    hb = batch.build_batch(records.AlignmentFile(path), o, mode="coordinate")
    exp, _ = oracle.collect(hb, p)
    assert exp.n == 2 and sorted(exp.type[:2].tolist()) == [0, 1]
    ins = int(np.nonzero(exp.type[:2] == 1)[0][0])
    assert exp.sequence(ins) == "ACGT" * 14 + "A"
    for filt in (0, 40):
        nb = NativeBam(path, threads=2)
        if filt:
            nb.set_seq_filter(filt)
        b, n = nb.read_batch(1 << 30, 20, "coordinate")
        A = nb.batch_arrays(b)
        assert n == 2 and int(A["cigar_off"][2] - A["cigar_off"][1]) == len(cig)
        assert np.array_equal(A["cigar"], hb.arrays["cigar"][:A["cigar"].size])
        got, _ = oracle.collect(b, p)
        assert got.first_difference(exp) is None
        if filt:
            assert A["seq"].size < 64                 # one insertion's worth of bases instead of ~50 kb
        nb.close()


def test_native_bam_reader_rewind_repeats_the_file(tmp_path):
    from svim_amd.bamio import NativeBam
    refs, lens = ["chr1", "chr2", "chr10"], [180000, 60000, 60000]
    recs = synth.coordinate_sort(synth.fuzz_split_reads(41, 90, refs, lens))
    path = str(tmp_path / "r.bam")
    records.write_bam(path, refs, lens, recs)
    nb = NativeBam(path, threads=3)
    nb.set_seq_filter(40)
    passes = []
    for it in range(3):
        if it:
            nb.rewind()
        got = []
        while True:
            b, n = nb.read_batch(50, 20, "coordinate")
            if n == 0:
                break
            A = nb.batch_arrays(b)
            got.append({k: A[k].copy() for k in ("flag", "pos", "read_id", "cigar", "seq", "seg_pos", "seq_rng_q0")})
        passes.append(got)
    nb.close()
    assert len(passes[0]) >= 3
    for other in passes[1:]:
        assert len(other) == len(passes[0])
        for x, y in zip(passes[0], other):
            for k in x:
                assert np.array_equal(x[k], y[k]), k


def _combine_case(engine_like):
    """g_combine's signatures -> (golden, options, our signature objects, the lazy 6-tuple) through `engine_like`.cluster"""
    g = H.load("g_combine.json.gz")
    o = H.options(g["options"])
    sigs = [H.row_sig(r) for r in g["signatures"]]
    tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(g["references"]))
    off, codes = convert.genome_arrays(o.genome, contigs.names)
    engine_like.set_genome(off, codes)
    ct = engine_like.cluster(_abi.Params.from_options(o), batch.contig_ranks(contigs.names), table=tab)
    return g, o, sigs, ct, contigs


def test_combine_consumers_on_lazy_cluster_lists(oracle, monkeypatch):
    """the COMBINE-side consumers (merge_translocations_at_insertions, flag_cutpaste_candidates, combine_clusters -
    src/svim/SVIM_merging.py:12-29,93-159, SVIM_COMBINE.py:332-478) on our lazy ClusterLists: expected rows from the reference's own functions
    (tests/golden/g_combine.json.gz); the consumer here replays their list / attribute protocol, including `del`, `extend` and `+`."""
    import combine_consumer as cc
    from svim_amd import _lib
    from svim_amd.lazy import ClusterList

    class LinkageOnly(object):                     # partition_and_cluster_candidates asks the process-wide engine for the linkage batch
        def linkage_fcluster(self, problems, cutoff):
            return [np.asarray(oracle.linkage_fcluster(n, np.asarray(d, dtype=np.float64), cutoff)) for n, d in problems]
    monkeypatch.setattr(_lib, "engine", lambda device=None: LinkageOnly())
    g, o, sigs, ct, contigs = _combine_case(oracle)
    H.compare_cluster_rows(H.cluster_rows(ct, contigs.names), g["clusters"])
    clusters = convert.cluster_objects(ct, sigs, contigs.names)
    assert all(isinstance(c, ClusterList) and c._objs is None for c in clusters)
    got = cc.consume(clusters, o, {id(s): i for i, s in enumerate(sigs)})
    diff = H.first_json_difference(got, g["expected"])
    assert diff is None, diff
    # the lists were mutated in place like the plain lists they stand for
    assert len(clusters[1]) == g["expected"]["n_ins_after"] and len(clusters[4]) == g["expected"]["n_dup_int_after"]
    assert len(clusters[5]) == g["expected"]["n_bnd_after_merge"]


def test_cluster_list_is_a_mutable_sequence(oracle):
    from svim_amd.lazy import ClusterList
    g, o, sigs, ct, contigs = _combine_case(oracle)
    dele, insr, inv, tan, dint, bnd = convert.cluster_objects(ct, sigs, contigs.names)
    n = len(insr)
    first, last = insr[0], insr[-1]
    del insr[0]
    assert len(insr) == n - 1 and insr[0] is not first and insr[-1] is last
    insr.extend(insr)                                                      # must terminate
    assert len(insr) == 2 * (n - 1)
    bnd.append(first)
    assert bnd[-1] is first and first in bnd
    joined = dele + inv
    assert isinstance(joined, list) and len(joined) == len(dele) + len(inv)
    dint.insert(0, last)
    assert dint[0] is last
    tan[0] = first
    assert tan[0] is first
    assert sorted(dele, key=lambda c: c.get_key()) == list(dele)
    with pytest.raises(TypeError):
        hash(dele)
    assert isinstance(reversed(dele).__next__(), type(dele[0]))
    import copy
    n_inv = len(inv)
    twin = copy.copy(inv)                                                  # built or not, a copy is a list of its own
    twin.append(first)
    assert len(inv) == n_inv and len(twin) == n_inv + 1 and twin[0] is inv[0]


def test_reader_keeps_the_last_batch_of_a_region_alive_across_a_seek(tmp_path):
    """a read that ends a region returns n = 0 - the two batch array sets must not alternate on it, or the first batch of the NEXT
    region (after svx_bam_seek) is decoded into the set that still holds the last batch of the region before, which the consumer may still be
    uploading.  Here the consumer is deliberately slow: it looks at a batch only after the reader has already produced the next one."""
    import threading
    import time
    from svim_amd import harness
    from svim_amd.bamio import NativeBam
    refs, lens = ["chr1", "chr2", "chr10", "chr3"], [100000, 80000, 80000, 60000]
    ref = synth.make_reference(61, list(zip(refs, lens)))
    recs = synth.coordinate_sort(synth.fuzz_split_reads(62, 200, refs, lens, max_sv_size=20000) +
                                 synth.planted_reads(63, 240, ref, refs, lens, n_sites=25, types=("DEL", "INS")))
    path = str(tmp_path / "regions.bam")
    records.write_bam(path, refs, lens, recs)
    bai = records.read_bai(path + ".bai")
    regions = [(bai[t][0], t) for t in (0, 2, 1, 3) if bai[t] is not None]          # four regions, visited out of file order
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5,
                   "partition_max_distance": 1000, "position_distance_normalizer": 900, "edit_distance_normalizer": 1.0, "cluster_max_distance": 0.5,
                   "all_bnds": False})

    def digest(nb, b):
        A = nb.batch_arrays(b)
        return tuple(int(A[k].astype(np.int64).sum()) for k in ("tid", "pos", "lseq", "read_id", "cigar", "seg_pos", "cigar_off")) + (int(b.n_rec),)
    # sequential reference: every batch looked at right away
    nb = NativeBam(path, threads=2)
    want = []
    for voff, last in regions:
        nb.seek(voff, last)
        while True:
            b, n = nb.read_batch(37, 20, "coordinate")
            if n == 0:
                break
            want.append(digest(nb, b))
    nb.close()

    class SlowEngine(object):
        def __init__(self):
            self.seen = []

        def accumulate(self, on):
            pass

        def set_slot_base(self, base):
            pass

        def collect(self, b, p, fetch=False):
            time.sleep(0.03)                         # the reader runs ahead: next read (and, at a region end, the n = 0 read + seek + read) happens now
            self.seen.append(digest(pipe.bam, b))
    eng = SlowEngine()
    pipe = harness.BamPipeline(path, o, eng, threads=2, batch_records=37, regions=regions, gpu_inflate=False, sparse_seq=False)
    n = pipe.run()
    pipe.bam.close()
    assert n == sum(w[-1] for w in want) and len(want) > 8
    assert eng.seen == want
    # a failing consumer: run() joins the reader before the error leaves (no thread left inside the handle that is about to be closed)
    class Failing(SlowEngine):
        def collect(self, b, p, fetch=False):
            raise RuntimeError("collect failed")
    pipe = harness.BamPipeline(path, o, Failing(), threads=2, batch_records=37, regions=regions, gpu_inflate=False, sparse_seq=False)
    before = threading.active_count()
    with pytest.raises(RuntimeError):
        pipe.run()
    pipe.bam.close()
    assert threading.active_count() <= before


def test_reader_verifies_block_crc_on_request(tmp_path, monkeypatch):
    """with SVX_BAM_VERIFY_CRC=1 the host reader checks the CRC32 of every BGZF block it inflates (htslib does): a block whose trailer
    CRC was altered is refused; without the switch the (still sound) DEFLATE stream is accepted."""
    import subprocess
    import sys
    from svim_amd._lib import SvxError
    refs, lens = ["chr1"], [200000]
    ref = synth.make_reference(5, list(zip(refs, lens)))
    recs = synth.coordinate_sort(synth.planted_reads(6, 300, ref, refs, lens, n_sites=10, types=("DEL", "INS")))
    good = str(tmp_path / "good.bam")
    records.write_bam(good, refs, lens, recs)
    raw = bytearray(open(good, "rb").read())
    # second block: flip one bit of its CRC32 (the 8 bytes before the next block header are CRC32 + ISIZE)
    bsize0 = raw[16] | (raw[17] << 8)
    at = bsize0 + 1
    bsize1 = raw[at + 16] | (raw[at + 17] << 8)
    raw[at + bsize1 + 1 - 8] ^= 1
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    code = ("import sys; sys.path.insert(0, %r)\nfrom svim_amd.bamio import NativeBam\nnb = NativeBam(sys.argv[1], threads=2)\nn = 0\n"
            "while True:\n    b, k = nb.read_batch(1000, 20, 'coordinate')\n    if k == 0: break\n    n += k\nprint('READ', n)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ)
    env.pop("SVX_BAM_VERIFY_CRC", None)
    out = subprocess.run([sys.executable, "-c", code, bad], capture_output=True, text=True, env=env, timeout=300)
    assert "READ %d" % len(recs) in out.stdout, out.stderr[-500:]
    env["SVX_BAM_VERIFY_CRC"] = "1"
    out = subprocess.run([sys.executable, "-c", code, good], capture_output=True, text=True, env=env, timeout=300)
    assert "READ %d" % len(recs) in out.stdout, out.stderr[-500:]
    out = subprocess.run([sys.executable, "-c", code, bad], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "CRC32" in out.stderr, (out.stdout, out.stderr[-500:])


def test_sample_bam_with_base_qualities_is_the_same_file_but_for_qual(tmp_path):
    """harness.write_bam_from_batch(qual_seed=...) - the literal-heavy sample of the inflate / end-to-end measurements: the records are the ones of the
    QUAL-less file (our Python BAM reader on both), only the quality bytes differ (Phred 1..50 instead of 0xff = absent), and the file deflates far worse."""
    import gzip
    from svim_amd import devsynth, harness
    b, genome, meta = devsynth.make_batch(n_reads=300, n50=3000, contig_len=400_000, seed=5, device="cpu")
    hb = b.slice_records(0, b.n_rec)
    p0, p1 = str(tmp_path / "plain.bam"), str(tmp_path / "qual.bam")
    n0, raw0 = harness.write_bam_from_batch(p0, hb, ["chr1"], [int(genome.numel())])
    n1, raw1 = harness.write_bam_from_batch(p1, hb, ["chr1"], [int(genome.numel())], qual_seed=3)
    assert n0 == n1 == b.n_rec and raw0 == raw1
    a0 = list(records.AlignmentFile(p0).fetch(until_eof=True))
    a1 = list(records.AlignmentFile(p1).fetch(until_eof=True))
    assert len(a0) == len(a1) == b.n_rec
    for x, y in zip(a0, a1):
        assert (x.query_name, x.flag, x.reference_id, x.reference_start, x.mapping_quality, x.cigartuples, x.query_sequence) == \
               (y.query_name, y.flag, y.reference_id, y.reference_start, y.mapping_quality, y.cigartuples, y.query_sequence)
    r0, r1 = gzip.open(p0).read(), gzip.open(p1).read()
    assert len(r0) == len(r1) == raw0
    diff = np.frombuffer(r0, dtype=np.uint8) != np.frombuffer(r1, dtype=np.uint8)
    q1 = np.frombuffer(r1, dtype=np.uint8)[diff]
    assert diff.any() and (np.frombuffer(r0, dtype=np.uint8)[diff] == 0xff).all() and q1.min() >= 1 and q1.max() <= 50
    import os
    assert os.path.getsize(p1) > 1.5 * os.path.getsize(p0)


def test_host_reader_on_damaged_files_under_sanitizers(tmp_path):
    """tools/bamio_fuzz.cpp: the host reader (csrc/bamio.cpp) built with AddressSanitizer + UndefinedBehaviorSanitizer reads damaged copies of a split-read BAM
    file to the end - record streams with overwritten bytes and extreme header fields inside well-formed BGZF blocks (so that the damage reaches the record
    decoder), cut streams, damaged compressed files; both sort modes, with and without the sparse-SEQ filter, 1-3 threads.  A read may be refused; it must
    not touch memory outside its buffers."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "bamio_fuzz")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                            os.path.join(repo, "svim_amd", "csrc", "bamio.cpp"), os.path.join(repo, "tools", "bamio_fuzz.cpp"), "-lz", "-lpthread", "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    refs, lens = ["chr1", "chr2", "chr10"], [100000, 80000, 60000]
    ref = synth.make_reference(61, list(zip(refs, lens)))
    recs = synth.coordinate_sort(synth.fuzz_split_reads(62, 200, refs, lens, max_sv_size=20000) + synth.planted_reads(63, 250, ref, refs, lens, n_sites=20, types=("DEL", "INS", "INV")))
    seed = str(tmp_path / "seed.bam")
    records.write_bam(seed, refs, lens, recs)
    run = subprocess.run([exe, seed, "160"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "seed: %d records in the stream, %d read" % (len(recs), len(recs)) in run.stdout
    tail = run.stdout.strip().splitlines()[-1]
    assert tail.startswith("damaged files: 160,") and int(tail.split("refused")[1]) > 50, tail
    # the undamaged file in windows of a few blocks, 1..6 threads, a rewind in the middle of a pass, the last pass in query-name mode: every pass returns
    # every record (the query-name peek across a chunk end once restored a position from before the window moved)
    for blocks in ("2", "3", "8"):
        env = dict(os.environ, SVX_BAM_CHUNK_BLOCKS=blocks)
        run = subprocess.run([exe, seed, "0"], capture_output=True, text=True, timeout=900, env=env)
        assert run.returncode == 0 and "threads 1..6 with rewinds: 0 problems" in run.stdout, (blocks, run.stdout[-500:], run.stderr[-2000:])
        assert "contig ranges out of file order: 0 problems" in run.stdout            # svx_bam_seek from every reference id's first record, twice, each range to its end
        # the HOST side of the device-resident reader (chunk slots rotating across seek / rewind, grow-and-retry, carry, both modes) over a CPU stand-in of the
        # decoder whose slot arrays are freed when a slot is loaded again: a batch read after its slot was reused would be a use-after-free
        assert "device reader's host side over the stand-in decoder: 0 problems" in run.stdout


def test_narrowing_staircase_window_rules_against_the_plain_dynamic_programme(tmp_path):
    """tools/stair_model.c: a cell-level model of d_edit_stair / stair_block (csrc/edit.hip) - the window's geometry, its boundary assumptions and the two cut-off
    rules that let it drop two words at the top / take none in at the bottom - against the plain dynamic programme: every ACCEPTED result (d <= kcap, the kernel's
    rule) is the edit distance, for any valid upper bound a pair may carry (exact, nearly exact, loose, useless), and narrowing never loses an answer the static
    window gives.  Related pairs with noise everywhere / bunched at either end, a deletion early paid back by an insertion late, shifted cores, low-complexity
    sequences, unequal lengths; windows of 4 .. 10 words.  (The GPU tests check the kernel itself against the oracle; this checks the RULES on far more cells.)"""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "stair_model")
    subprocess.check_call(["gcc", "-O2", os.path.join(repo, "tools", "stair_model.c"), "-o", exe])
    run = subprocess.run([exe, "2500", "11"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0 and " 0 wrong" in run.stdout, (run.stdout[-500:], run.stderr[-2000:])
    acc = int(run.stdout.split("settings:")[1].split("accepted")[0])
    assert acc > 8000, run.stdout


def test_keyboard_interrupt_ends_the_reading_and_keeps_what_was_collected(tmp_path):
    """src/svim/SVIM_COLLECT.py:126-128,164-166: Ctrl-C breaks the loop over the alignments and the function returns the signatures collected so far - the
    pipeline goes on with them.  Here: (1) the Python batcher stops reading where the interrupt falls (query-name order: the incomplete last read group is
    dropped - the reference only ever processes groups its iterator has completed, :8-41); (2) harness.BamPipeline ends its pass with the batches whose
    COLLECT has completed, leaves libsvx's reader thread cleanly and reports the records of those batches."""
    from svim_amd import harness
    refs, lens = ["chr1", "chr2", "chr10"], [180000, 60000, 60000]
    recs = synth.fuzz_split_reads(43, 60, refs, lens)                       # query-name order: the records of a read are adjacent

    class Interrupting(object):
        references = refs

        def __init__(self, recs, after):
            self.recs, self.after = recs, after

        def fetch(self, until_eof=True):
            for k, a in enumerate(self.recs):
                if k == self.after:
                    raise KeyboardInterrupt()
                yield a

        def get_tid(self, name):
            return refs.index(name) if name in refs else -1

    o = H.options({})
    names = [a.query_name for a in recs]
    cut = next(k for k in range(40, len(recs)) if names[k] == names[k - 1])          # inside a read group
    hb = batch.build_batch(Interrupting(recs, cut), o, mode="queryname")
    first_of_group = next(k for k in range(cut, -1, -1) if names[k] != names[cut - 1]) + 1
    assert hb.n_rec == first_of_group < cut                                          # whole groups only
    srt = synth.coordinate_sort(recs)
    hb = batch.build_batch(Interrupting(srt, 37), o, mode="coordinate")
    assert hb.n_rec == 37 and len(hb.read_names) > 5
    whole = batch.build_batch(Interrupting(srt, 10 ** 9), o, mode="coordinate")
    assert whole.n_rec == len(srt)

    # (2) the native pipeline: a stand-in engine (no GPU here) whose second svx_collect is where the interrupt arrives
    path = str(tmp_path / "ki.bam")
    records.write_bam(path, refs, lens, srt)

    class Eng(object):
        device = None

        def __init__(self):
            self.calls, self.acc, self.sizes = 0, [], []

        def accumulate(self, on):
            self.acc.append(bool(on))

        def set_slot_base(self, b):
            pass

        def collect(self, b, p, fetch=False):
            # an interrupt that arrives while a thread is inside a foreign (ctypes) call is raised when the call RETURNS: the batch has been
            # appended to the accumulated lists by then - the stand-in records it and raises
            self.calls += 1
            self.sizes.append(int(b.n_rec))
            if self.calls == 2:
                raise KeyboardInterrupt()

    eng = Eng()
    pipe = harness.BamPipeline(path, o, eng, threads=2, batch_records=40, mode="coordinate", gpu_inflate=False, device_decode=False)
    n = pipe.run()                                                                   # does not raise
    # the counters agree with what CLUSTER will see: both batches whose svx_collect returned, none of the batch that was being read
    assert pipe.interrupted and n == 80 and eng.sizes == [40, 40] and pipe.stats["batches"] == 2
    assert sum(c for _, c in pipe.region_slots) == 80
    pipe.close()
    assert eng.acc == [True, False]
    # an uninterrupted pass over the same file for comparison
    eng2 = Eng()
    eng2.calls = 10
    pipe = harness.BamPipeline(path, o, eng2, threads=2, batch_records=40, mode="coordinate", gpu_inflate=False, device_decode=False)
    assert pipe.run() == len(srt) and not pipe.interrupted and sum(eng2.sizes) == len(srt)
    pipe.close()


def test_gpu_suite_order_puts_hot_path_and_configs_in_front_of_the_readers():
    """tests/conftest.py orders the `-m gpu` suite: hot-path parity, then the BASELINE configs at scale (configs[1] at its full size first), then the rank
    exchange, then everything that reads BAM files - so that `pytest -x` can never again stop in a reader test before the configs have run."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(repo, "tests"), "-m", "gpu", "--collect-only", "-q"], capture_output=True, text=True, cwd=repo)
    ids = [l.strip() for l in out.stdout.splitlines() if "::" in l]
    assert len(ids) > 120
    import conftest
    tiers = [conftest.gpu_tier(i) for i in ids]
    assert tiers == sorted(tiers), [(i, t) for i, t in zip(ids, tiers)][:5]
    first_reader = tiers.index(3)
    names = [i.split("::")[-1] for i in ids]
    for must in ("test_c1_full_bench_size_vs_oracle_and_properties", "test_c2_hifi_full_sv_type_set_vs_oracle", "test_c4_clr_partition_max_distance_sweep_vs_oracle[1000]",
                 "test_c1_bench_shape_sample_vs_reference_golden", "test_collect_golden_and_oracle[0]", "test_cluster_golden_and_oracle[0]", "test_edit_distance_golden"):
        assert names.index(must) < first_reader, must
    assert names.index("test_c1_full_bench_size_vs_oracle_and_properties") == tiers.index(1)          # configs[1] at the bench size opens the configs
    for late in ("test_device_bam_decode_equals_host_reader[None]", "test_reader_life_cycle_stress_with_pageable_copies[1]", "test_c3_whole_genome_contig_sharded_ranks_on_one_gpu[8]"):
        assert names.index(late) >= first_reader, late


def test_bounce_buffer_copy_layer_over_an_asynchronous_model(tmp_path):
    """svim_amd/csrc/hostcopy.hip - the page-locked bounce buffers EVERY host <-> device copy of the library goes through since round 5 - compiled
    for the host over a model of the HIP calls it makes in which copies really are asynchronous (tools/hostcopy_model.cpp: bytes move only when a stream is drained, an
    event is waited for or polled): uploads whose source is overwritten the moment h2d returns, fetches that must hold the uploaded bytes after finish(), sizes around
    every boundary of the layer (64 KiB slots, 8 MiB pieces, the four-thread upload of >= 64 MB), a HostCopy dropped with copies in flight, several threads sharing the
    slot pool.  A slot handed out again before the copy out of it has run, or a destination handed over early, is wrong data; under AddressSanitizer + UBSan, and under
    ThreadSanitizer (two threads writing one slot)."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("no HIP headers")
    for san, args, env in (("address,undefined", ["4", "30"], {"ASAN_OPTIONS": "detect_leaks=0"}),       # (the slot pool lives as long as the process, by design)
                           ("thread", ["3", "8"], {})):
        exe = str(tmp_path / ("hostcopy_model_" + san[:3]))
        build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-x", "c++",
                                os.path.join(repo, "svim_amd", "csrc", "hostcopy.hip"), os.path.join(repo, "tools", "hostcopy_model.cpp"), "-lpthread", "-o", exe],
                               capture_output=True, text=True)
        if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
            pytest.skip("no sanitizer runtime in this toolchain")
        assert build.returncode == 0, build.stderr[-2000:]
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert run.returncode == 0 and "every byte arrived" in run.stdout and "runtime error" not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr, (san, run.stdout[-500:], run.stderr[-3000:])
