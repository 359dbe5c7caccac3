"""BAM files as OTHER programs write them (tests/foreign_bam.py): the host reader (csrc/bamio.cpp), the Python batcher and - on the GPU - the device-resident
reader (csrc/bamdev.hip) must hand out the same record batches; COLLECT on them equals the oracle's.  What the reference gets from pysam / htslib
(src/svim/SVIM_COLLECT.py:133,142-143)."""
import random
import zlib

import numpy as np
import pytest

import foreign_bam as FB
import helpers as H
from svim_amd import _abi, batch, records, synth

REFS, LENS = ["chr1", "chr2", "chr10"], [180000, 60000, 60000]

LAYOUTS = {
    # name: (layout, deflate per block, empty_every, block_payload)
    "htslib": ("htslib", ((6, zlib.Z_DEFAULT_STRATEGY),), 0, 0xff00),
    "htslib_level1": ("htslib", ((1, zlib.Z_DEFAULT_STRATEGY),), 0, 0xff00),
    "htslib_stored": ("htslib", ((0, zlib.Z_DEFAULT_STRATEGY),), 0, 0xff00),
    "htslib_fixed_huffman": ("htslib", ((6, zlib.Z_FIXED),), 0, 0xff00),
    "mixed_blocks_with_eof_markers": ("htslib", ((6, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (9, zlib.Z_FILTERED),
                                                 (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)), 3, 0xff00),
    "flat_small_blocks": ("flat", ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY)), 5, 3001),
}


def foreign_case(seed, with_long=True):
    """records of every kind the readers special-case, decorated with aux fields of every type; returns (AlignedSegments in file order, record bytes)"""
    rng = random.Random(seed)
    ref = synth.make_reference(31, list(zip(REFS, LENS)))
    recs = synth.fuzz_split_reads(seed + 1, 160, REFS, LENS) + synth.planted_reads(seed + 2, 220, ref, REFS, LENS, n_sites=24, types=("DEL", "INS", "INV"))
    recs = synth.coordinate_sort(recs)
    # names of 1 and 254 characters (whole reads: every record of the read is renamed)
    names = sorted({a.query_name for a in recs})
    rename = {names[3]: "q", names[7]: "L" * 254, names[11]: "r" * 253}
    for a in recs:
        a.query_name = rename.get(a.query_name, a.query_name)
    # SEQ '*': secondary alignments (skipped by COLLECT) and a few primaries whose CIGAR holds no reportable insertion
    n_star = 0
    for a in recs:
        big_ins = any(o == 1 and l >= 40 for o, l in a._cigar)
        if (a.flag & 256) or (not big_ins and "SA" not in a._tags and rng.random() < 0.06):
            a._seq = ""
            n_star += 1
    assert n_star >= 5
    out = []
    for k, a in enumerate(recs):
        qual = bytes(rng.randrange(0, 60) for _ in range(len(a._seq or ""))) if k % 3 else None
        out.append(FB.record_bytes(a, FB.decorate(rng, a, k), qual))
    if with_long:
        # one record longer than two BGZF blocks (a 150 kb read) and the > 65535-operation CIGAR that lives in the CG tag, fields on both sides of it
        short, long_rec, cig = H.long_cigar_records()
        long_rec.reference_id, short.reference_id = 2, 2
        big = records.AlignedSegment()
        big.query_name, big.flag, big.reference_id, big.reference_start, big._mapq = "big150k", 0, 2, 100, 60
        big._cigar = [(0, 70000), (1, 55), (0, 79945)]
        big._seq = "".join(rng.choice("ACGT") for _ in range(150000))
        big._tags = {}
        big.next_reference_id, big.next_reference_start, big.template_length = -1, -1, 0
        recs = [r for r in recs]
        tail = [(big, FB.decorate(rng, big, 1), bytes(rng.randrange(0, 60) for _ in range(150000))),
                (long_rec, [("NM", "i", 7), ("CG", "B", ("I", [(l << 4) | o for o, l in long_rec._cigar])), ("tp", "A", "P"), ("ML", "B", ("C", [1, 2, 3]))], None)]
        # the same long CIGAR as another writer might store it: placeholder <l_seq>S<ref_len>M, CG of sub-type 'i' (htslib accepts both)
        short2, long2, _ = H.long_cigar_records()
        long2.reference_id, long2.query_name = 2, "long_other_writer"
        tail.append((long2, [("CG", "B", ("i", [(l << 4) | o for o, l in long2._cigar])), ("NM", "i", 3)], None))
        # htslib's bam_tag2cigar in full (what pysam hands the reference): the placeholder stays when the CG array is SHORTER than the placeholder CIGAR, and
        # bam_aux_get finds the FIRST field named CG - of another type it hides a later array, of the right type it wins over a later one
        def small(name, cigar):
            a = records.AlignedSegment()
            a.query_name, a.flag, a.reference_id, a.reference_start, a._mapq = name, 0, 2, 100, 60
            a._cigar, a._seq, a._tags = list(cigar), "".join(rng.choice("ACGT") for _ in range(50)), {}
            a.next_reference_id, a.next_reference_start, a.template_length = -1, -1, 0
            return a
        real_a, real_b = [(0, 30), (2, 60), (0, 20)], [(0, 10), (1, 45), (0, 5)]
        pk = lambda cg: [(l << 4) | o for o, l in cg]                                   # noqa: E731
        tail.append((small("cg_shorter_than_placeholder", [(4, 50), (3, 300), (3, 200)]), [("NM", "i", 1), ("CG", "B", ("I", pk([(0, 50), (2, 10)])))], None))
        tail.append((small("cg_first_of_other_type", [(4, 50), (3, 500)]), [("CG", "Z", "not an array"), ("CG", "B", ("I", pk(real_a)))], None))
        two = small("cg_two_arrays", real_a)                                             # (the truth list holds the CIGAR the FIRST array restores)
        tail.append((two, [("tp", "A", "P"), ("CG", "B", ("I", pk(real_a))), ("CG", "B", ("I", pk(real_b)))], None))
        # the same rule for SA (pysam's get_tag is bam_aux_get too): of two SA fields only the FIRST is ever looked at (the native readers took the last)
        twice = small("sa_twice", [(0, 30), (4, 20)])
        twice._tags = {"SA": "%s,2001,+,30S20M,60,0;" % REFS[0]}
        tail.append((twice, [("NM", "i", 0), ("SA", "Z", twice._tags["SA"]), ("SA", "Z", "%s,7001,-,25S25M,50,1;%s,9001,+,10M40S,33,2;" % (REFS[1], REFS[0]))], None))
        # keep the file coordinate-sorted: all go behind the last chr10 record
        last_pos = max([a.reference_start for a in recs if a.reference_id == 2] + [0])
        for k, (a, items, q) in enumerate(tail):
            a.reference_start = last_pos + 10
            last_pos += 10
            recs.append(a)
            if a.query_name == "cg_two_arrays":                                          # written behind a placeholder although it is short
                ph = small(a.query_name, [(4, 50), (3, 110)])
                ph._seq, ph.reference_start = a._seq, a.reference_start
                out.append(FB.record_bytes(ph, items, q))
            else:
                out.append(FB.record_bytes(a, items, q, placeholder_op=0 if k == 2 else 3))
    return recs, out


def _python_batch(path, mode="coordinate"):
    return batch.build_batch(records.AlignmentFile(path), H.options({"min_mapq": 20}), mode=mode)


def _assert_same_batch(A, hb, what):
    for k in _abi.BATCH_DTYPES:
        exp, got = hb.arrays[k], A[k]
        assert np.array_equal(got, exp[:got.size]), (what, k)


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_host_reader_on_foreign_bam_files(tmp_path, layout):
    from svim_amd.bamio import NativeBam
    recs, rb = foreign_case(5)
    lay, deflate, empty_every, payload = LAYOUTS[layout]
    path = str(tmp_path / "f.bam")
    n_blocks = FB.write(path, REFS, LENS, rb, layout=lay, deflate=deflate, empty_every=empty_every, block_payload=payload, tids=[a.reference_id for a in recs])
    assert n_blocks > 6
    # the Python reader sees what was written: names, flags, CIGARs (the CG one restored), every aux field parsed
    back = list(records.AlignmentFile(path).fetch(until_eof=True))
    assert [a.query_name for a in back] == [a.query_name for a in recs]
    assert [a._cigar for a in back] == [a._cigar for a in recs]
    assert all("ML" in a._tags and "XH" in a._tags for a in back[:50])
    hb = _python_batch(path)
    nb = NativeBam(path, threads=3)
    b, n = nb.read_batch(1 << 30, 20, "coordinate")
    assert n == hb.n_rec == len(recs) and b.n_seg == hb.n_seg
    _assert_same_batch(nb.batch_arrays(b), hb, layout)
    assert nb.read_names() == hb.read_names
    nb.close()
    # in small batches, with CRC verification on
    import os
    os.environ["SVX_BAM_VERIFY_CRC"] = "1"
    try:
        nb = NativeBam(path, threads=2)
        tot, flags = 0, []
        while True:
            b, n = nb.read_batch(41, 20, "coordinate")
            if n == 0:
                break
            flags.append(nb.batch_arrays(b)["flag"])
            tot += n
        nb.close()
    finally:
        del os.environ["SVX_BAM_VERIFY_CRC"]
    assert tot == hb.n_rec and np.array_equal(np.concatenate(flags) & 0xfff, hb.arrays["flag"] & 0xfff)


def test_oracle_collect_on_a_foreign_bam(tmp_path, oracle):
    """COLLECT over the host reader's batch of a foreign file = COLLECT over the Python batcher's (same signatures, in order)"""
    from svim_amd.bamio import NativeBam
    recs, rb = foreign_case(9)
    path = str(tmp_path / "f.bam")
    FB.write(path, REFS, LENS, rb, tids=[a.reference_id for a in recs])
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5, "all_bnds": True})
    p = _abi.Params.from_options(o)
    hb = batch.build_batch(records.AlignmentFile(path), o, mode="coordinate")
    sig, bnd = oracle.collect(hb, p)
    assert sig.n > 100
    nb = NativeBam(path, threads=2)
    b, n = nb.read_batch(1 << 30, 20, "coordinate")
    sig2, bnd2 = oracle.collect(b, p)
    nb.close()
    assert sig.first_difference(sig2) is None and bnd.first_difference(bnd2) is None


@pytest.mark.gpu
@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_device_reader_on_foreign_bam_files(tmp_path, layout, monkeypatch):
    """device-resident reader == host reader == Python batcher on every foreign layout (whole file in one chunk, and in chunks of a few blocks: records,
    record headers and the oversize record straddle chunk boundaries), then svx_collect on the device batch == the oracle"""
    from svim_amd.bamio import NativeBam
    recs, rb = foreign_case(5)
    lay, deflate, empty_every, payload = LAYOUTS[layout]
    path = str(tmp_path / "f.bam")
    FB.write(path, REFS, LENS, rb, layout=lay, deflate=deflate, empty_every=empty_every, block_payload=payload, tids=[a.reference_id for a in recs])
    hb = _python_batch(path)
    for chunk_blocks in (None, "4"):
        if chunk_blocks:
            monkeypatch.setenv("SVX_BAM_DEV_CHUNK_BLOCKS", chunk_blocks)
        else:
            monkeypatch.delenv("SVX_BAM_DEV_CHUNK_BLOCKS", raising=False)
        dev = NativeBam(path, threads=2)
        dev.set_device_decode(0)
        rows, names = [], []
        while True:
            b, n = dev.read_batch(97, 20, "coordinate")
            if n == 0:
                break
            A = dev.batch_arrays(b)
            nm = dev.read_names()
            names += [nm[int(i)] for i in A["read_id"]]
            rows.append(A)
        dev.close()
        got = H.concat_batch_rows(rows)
        want = H.concat_batch_rows([{k: hb.arrays[k] for k in hb.arrays}])
        assert len(got) == len(want) == len(recs), (layout, chunk_blocks, len(got), len(want))
        bad = [i for i, (g, w) in enumerate(zip(got, want)) if g != w]
        assert not bad, (layout, chunk_blocks, bad[:5])
        assert names == [hb.read_names[int(i)] for i in hb.arrays["read_id"][:hb.n_rec]]


@pytest.mark.gpu
def test_collect_on_the_device_batch_of_a_foreign_bam(tmp_path, oracle):
    from svim_amd import _lib, harness
    recs, rb = foreign_case(9)
    path = str(tmp_path / "f.bam")
    FB.write(path, REFS, LENS, rb, tids=[a.reference_id for a in recs], deflate=((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)), empty_every=4)
    o = H.options({"min_mapq": 20, "min_sv_size": 40, "max_sv_size": 100000, "segment_gap_tolerance": 10, "segment_overlap_tolerance": 5, "all_bnds": True,
                   "partition_max_distance": 1000, "position_distance_normalizer": 900, "edit_distance_normalizer": 1.0, "cluster_max_distance": 0.5})
    p = _abi.Params.from_options(o)
    hb = batch.build_batch(records.AlignmentFile(path), o, mode="coordinate")
    osig, obnd = oracle.collect(hb, p)
    eng = _lib.Engine(0)
    try:
        pipe = harness.BamPipeline(path, o, eng, threads=2, batch_records=150)
        assert pipe.run() == len(recs) and pipe.stats["batches"] >= 3
        sig = eng.fetch_signatures(0)
        bnd = eng.fetch_signatures(1)
        names = pipe.bam.read_names()
        pipe.close()
    finally:
        eng.close()
    assert H.table_rows(sig, REFS, names) == H.table_rows(osig, REFS, hb.read_names)
    assert H.table_rows(bnd, REFS, names) == H.table_rows(obnd, REFS, hb.read_names)
