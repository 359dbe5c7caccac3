"""Pin the oracle (oracle/svx_oracle.c) against golden vectors produced by running the reference
(tests/golden/make_golden.py).  CPU only."""
import struct

import numpy as np
import pytest

import helpers as H
from svim_amd import _abi, convert, batch


def test_g1_cigar_indel(oracle):
    g = H.load("g1_cigar_indel.json.gz")
    for c in g["cases"]:
        got = oracle.cigar_indel([tuple(t) for t in c["tuples"]], c["min_length"])
        assert got == [tuple(x) for x in c["expect"]]


def test_g8_rng(oracle):
    g = H.load("g8_rng.json.gz")["rng"]
    for k, vals in g["getrandbits"].items():
        got = oracle.getrandbits(1524, int(k), len(vals))
        assert got.tolist() == vals
    ns = [s["n"] for s in g["samples"]]
    got = oracle.sample_sequence(1524, ns)
    for row, s in zip(got, g["samples"]):
        assert row.tolist() == s["idx"]


def test_edit_distance(oracle):
    g = H.load("g_editdistance.json.gz")
    for a, b, d in g["cases"]:
        assert oracle.edit_distance(a, b) == d
        assert oracle.edit_distance(b, a) == d


def test_linkage_fcluster(oracle):
    g = H.load("g_linkage.json.gz")
    for c in g["cases"]:
        d = np.array([float.fromhex(x) for x in c["d"]])
        lab, Z = oracle.linkage_fcluster(c["n"], d, c["t"], want_z=True)
        Zexp = np.array([[float.fromhex(v) for v in row] for row in c["Z"]])
        assert np.array_equal(Z[:c["n"] - 1], Zexp)
        assert lab.tolist() == c["labels"]


def test_g3_satag_rebuild():
    """SA-tag reconstruction == the supplementary records themselves (src/tests/test_satag.py:21-34), through the
    host batcher + the geometry rules the device applies."""
    g = H.load("g3_satag.json")
    assert len(g["sa_rebuilt"]) == 3
    for a, b in zip(g["sa_rebuilt"], g["supplementary_records"]):
        assert a == b


@pytest.mark.parametrize("idx", range(len(H.load("g2_collect.json.gz")["cases"])))
def test_g2_collect(oracle, idx):
    g = H.load("g2_collect.json.gz")
    if idx >= len(g["cases"]):
        pytest.skip("no such case")
    case = g["cases"][idx]
    bam, hb, o = H.sam_case_batch(case, g)
    sig, bnd = oracle.collect(hb, _abi.Params.from_options(o))
    assert H.table_rows(sig, hb.references, hb.read_names) == case["signatures"]
    assert H.table_rows(bnd, hb.references, hb.read_names) == case["bnds"]


def test_g4_partitions(oracle):
    g4 = H.load("g4_partitions.json.gz")
    g5 = H.load("g5_cluster.json.gz")
    cases = {c["name"]: c for c in g5["cases"]}
    for p in g4["partitions"]:
        case = cases[p["case"]]
        sigs = [H.row_sig(r) for r in case["signatures"]]
        tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(H.REFS))
        code = _abi.TYPE_CODE[p["type"]]
        sidx, pid = oracle.form_partitions(tab, batch.contig_ranks(contigs.names), case["options"]["partition_max_distance"])
        sel = tab.type[sidx] == code
        got = {}
        for i, q in zip(sidx[sel], pid[sel]):
            got.setdefault(int(q), []).append(int(i))
        assert [got[k] for k in sorted(got)] == p["partitions"]


def test_g6_distance(oracle):
    g = H.load("g6_distance.json.gz")
    sigs = [H.row_sig(r) for r in g["signatures"]]
    tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(H.REFS))
    off, codes = convert.genome_arrays(H.options({}).genome, contigs.names)
    oracle.set_genome(off, codes)
    p = _abi.Params.from_options(H.options({}))
    for i, j, hexd in g["pairs"]:
        d = oracle.span_position_distance(tab, i, j, p)
        assert struct.pack("<d", d).hex() == hexd, (i, j, g["signatures"][i][0])


@pytest.mark.parametrize("idx", range(len(H.load("g5_cluster.json.gz")["cases"])))
def test_g5_cluster(oracle, idx):
    g = H.load("g5_cluster.json.gz")
    if idx >= len(g["cases"]):
        pytest.skip("no such case")
    case = g["cases"][idx]
    o = H.options(case["options"])
    sigs = [H.row_sig(r) for r in case["signatures"]]
    tab, contigs, reads = convert.sigtable_from_objects(sigs, convert.Interner(H.REFS))
    off, codes = convert.genome_arrays(o.genome, contigs.names)
    oracle.set_genome(off, codes)
    ct = oracle.cluster(_abi.Params.from_options(o), batch.contig_ranks(contigs.names), table=tab)
    H.compare_cluster_rows(H.cluster_rows(ct, contigs.names), case["clusters"])


def test_c1_config0_end_to_end(oracle):
    """BASELINE.json configs[0] (10k-read synthetic 1-contig input, DEL/INS only): the oracle reproduces what the
    reference's CPU path returned for COLLECT and CLUSTER."""
    from svim_amd import synth
    g, refs, recs = H.c1_case()
    bam = H.records.AlignmentFile(text=synth.sam_text(["chr1"], [2000000], recs))
    o = H.options(g["options"])
    hb = batch.build_batch(bam, o, mode="coordinate")
    p = _abi.Params.from_options(o)
    sig, bnd = oracle.collect(hb, p)
    assert H.table_rows(sig, hb.references, hb.read_names) == g["signatures"]
    off, codes = convert.genome_arrays(refs, ["chr1"])
    oracle.set_genome(off, codes)
    ct = oracle.cluster(p, hb.contig_rank, source=0)
    H.compare_cluster_rows(H.cluster_rows(ct, ["chr1"]), g["clusters"])


def test_c1_config0_full_size_end_to_end(oracle):
    """BASELINE.json configs[0] at the size SURVEY.md section 8(d) C1 states (250 Mb contig, reads ~ triangular(100, 20000, 15000), 1.2*10^7 CIGAR
    operations): the oracle reproduces what the reference's CPU path returned (tests/golden/g_c1_full.json.gz)."""
    g = H.load("g_c1_full.json.gz")
    hb, genome, meta = H.c1_full_case()
    assert hb.n_rec == g["n_records"] and int(meta["n_ops"]) == g["n_ops"]
    names = ["r%08d" % i for i in range(int(hb.arrays["read_id"].max()) + 1)]
    p = _abi.Params.from_options(H.options(g["options"]))
    sig, bnd = oracle.collect(hb, p)
    assert H.table_rows(sig, ["chr1"], names) == g["signatures"] and bnd.n == g["n_bnds"]
    oracle.set_genome(np.array([0, genome.shape[0]], dtype=np.int64), genome)
    ct = oracle.cluster(p, hb.contig_rank, source=0)
    H.compare_cluster_rows(H.cluster_rows(ct, ["chr1"]), g["clusters"])


def test_threaded_pair_distances_do_not_change_the_oracle():
    """svo_set_threads (used by the full-size GPU parity test so that the checker finishes in a minute): the pair distances of svo_cluster on
    worker threads - the tables stay those of the single-threaded run AND of the reference (configs[0] golden, all of its partitions)."""
    from oracle import oracle as om
    from svim_amd import synth
    g, refs, recs = H.c1_case()
    bam = H.records.AlignmentFile(text=synth.sam_text(["chr1"], [2000000], recs))
    o = H.options(g["options"])
    hb = batch.build_batch(bam, o, mode="coordinate")
    p = _abi.Params.from_options(o)
    off, codes = convert.genome_arrays(refs, ["chr1"])
    tables = []
    for threads in (1, 5, 16):
        oc = om.Oracle()
        oc.set_threads(threads)
        oc.set_genome(off, codes)
        oc.collect(hb, p)
        ct = oc.cluster(p, hb.contig_rank, source=0)
        tables.append((ct, oc.stats()))
    H.compare_cluster_rows(H.cluster_rows(tables[1][0], ["chr1"]), g["clusters"])
    for ct, st in tables[1:]:
        assert tables[0][0].first_difference(ct, rtol=0.0) is None
        assert st["n_edit_pairs"] == tables[0][1]["n_edit_pairs"] and st["n_pairs"] == tables[0][1]["n_pairs"]


def test_c1_bench_shape_sample_oracle_vs_reference(oracle):
    """the oracle on the reference's outputs at the bench workload's shape (g_c1_bench_sample: configs[1]'s densities, 20 000 reads + supplementary records,
    INV included) - the fixture the GPU is held to in tests/test_gpu_workloads.py"""
    g = H.load("g_c1_bench_sample.json.gz")
    hb, genome, meta = H.c1_bench_sample_case()
    assert hb.n_rec == g["n_records"]
    import types
    o = types.SimpleNamespace(**g["options"])
    p = _abi.Params.from_options(o)
    oracle.set_genome(np.array([0, genome.shape[0]], dtype=np.int64), genome)
    oracle.set_threads(H.granted_cpus())
    try:
        sig, bnd = oracle.collect(hb, p)
        names = ["r%08d" % i for i in range(int(hb.arrays["read_id"].max()) + 1)]
        assert H.table_rows(sig, ["chr1"], names) == g["signatures"] and bnd.n == g["n_bnds"]
        ct = oracle.cluster(p, hb.contig_rank, source=0)
    finally:
        oracle.set_threads(1)
    H.compare_cluster_rows(H.cluster_rows(ct, ["chr1"]), g["clusters"])
