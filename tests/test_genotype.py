"""GENOTYPE: svim_amd.SVIM_genotyping.genotype against the golden vectors produced by the reference's
genotype() (tests/golden/make_golden.py:gen_genotype) - the C oracle on the CPU, the HIP interval join with -m gpu."""
import types

import numpy as np
import pytest

from svim_amd import SVIM_genotyping, records, synth
from tests import helpers as H


class _Sig(object):
    def __init__(self, read):
        self.read = read


class _Candidate(object):
    """Quacks like the reference's candidates (src/svim/SVCandidate.py): get_source / get_destination, score, members."""

    def __init__(self, typ, contig, start, end, members, score):
        self.type, self.locus, self.members, self.score = typ, (contig, start, end), [_Sig(m) for m in members], score
        self.support_fraction, self.genotype, self.ref_reads, self.alt_reads = ".", "./.", None, None

    def get_source(self):
        return self.locus if self.type in ("DEL", "INV") else ("chr1", 100, 100 + self.locus[2] - self.locus[1])

    def get_destination(self):
        return self.locus


@pytest.fixture(scope="module")
def golden():
    g = H.load("g_genotype.json.gz")
    bam = records.AlignmentFile(text=synth.genotype_sam_text(g["references"], g["lengths"], g["rows"]))
    return g, bam


def _run(g, bam, engine):
    o = types.SimpleNamespace(**g["options"])
    out = {}
    for case in g["cases"]:
        cands = [_Candidate(case["type"], r[0], r[1], r[2], r[3], r[4]) for r in case["candidates"]]
        SVIM_genotyping.genotype(cands, bam, case["type"], o, engine=engine)
        out[case["type"]] = [[c.support_fraction, c.genotype, c.ref_reads, c.alt_reads] for c in cands]
    return out


def _check(got, g):
    for case in g["cases"]:
        for i, (a, e) in enumerate(zip(got[case["type"]], case["expected"])):
            assert a[1:] == e[1:], (case["type"], i, a, e)
            assert a[0] == e[0] or abs(a[0] - e[0]) < 1e-15, (case["type"], i, a, e)


def test_rows_regenerate(golden):
    g, _ = golden
    assert synth.genotype_rows(5, g["lengths"]) == g["rows"]


def test_region_fetch_follows_htslib_overlap_rule(golden):
    g, bam = golden
    got = [a.query_name for a in bam.fetch(contig="chr2", start=1000, stop=1500)]
    # bam_endpos: pos + reference length, pos + 1 for an unmapped-but-placed record
    exp = [r[0] for r in g["rows"] if r[2] == 1 and r[3] < 1500 and (r[3] + 1 if r[1] & 4 else r[3] + r[5]) > 1000]
    assert got == exp and len(got) > 3
    assert bam.get_reference_length("chr2") == g["lengths"][1]


def test_oracle_genotype_matches_reference(golden, oracle):
    g, bam = golden
    _check(_run(g, bam, oracle), g)


def test_span_position_distance_matches_reference_formula():
    c = _Candidate("DEL", "chr1", 1000, 1500, [], 5)
    s = types.SimpleNamespace(type="DEL", get_source=lambda: ("chr1", 1010, 1490), get_destination=lambda: None)
    d = SVIM_genotyping.span_position_distance(c, s, 900)
    assert d == min(10, 10, 0) / 900 + 20 / 500
    s2 = types.SimpleNamespace(type="INV", get_source=lambda: ("chr1", 1010, 1490), get_destination=lambda: None)
    assert SVIM_genotyping.span_position_distance(c, s2, 900) == float("inf")


@pytest.mark.gpu
def test_gpu_genotype_matches_reference_and_oracle(golden, oracle):
    from svim_amd._lib import engine
    g, bam = golden
    eng = engine()
    got = _run(g, bam, eng)
    _check(got, g)
    assert got == _run(g, bam, oracle)


@pytest.mark.gpu
def test_gpu_genotype_random_vs_oracle(oracle):
    """Bigger random case (deeper piles, three contigs, empty member lists, candidates at the contig ends)."""
    import random
    from svim_amd._lib import engine
    references, lengths = ["a", "b", "c"], [90000, 500, 40000]
    rows = synth.genotype_rows(21, lengths, n_reads=9000, hot=((0, 45000, 2500), (2, 100, 900), (2, 39900, 900)))
    bam = records.AlignmentFile(text=synth.genotype_sam_text(references, lengths, rows))
    rng = random.Random(4)
    names = [r[0] for r in rows]
    o = types.SimpleNamespace(minimum_score=0, minimum_depth=4, homozygous_threshold=0.8, heterozygous_threshold=0.2, min_mapq=20)
    eng = engine()
    for typ in ("DEL", "INS"):
        cands_a, cands_b = [], []
        for _ in range(400):
            tid = rng.randrange(3)
            span = min(rng.choice((1, 60, 500, 3999, 4001, 20000)), lengths[tid] - 1)
            start = rng.choice((0, lengths[tid] - span - 1, rng.randint(0, lengths[tid] - span - 1)))
            members = rng.sample(names, rng.choice((0, 0, 5, 50, 400)))
            for lst in (cands_a, cands_b):
                lst.append(_Candidate(typ, references[tid], start, start + span, members, 10))
        SVIM_genotyping.genotype(cands_a, bam, typ, o, engine=eng)
        SVIM_genotyping.genotype(cands_b, bam, typ, o, engine=oracle)
        a = [[c.support_fraction, c.genotype, c.ref_reads, c.alt_reads] for c in cands_a]
        b = [[c.support_fraction, c.genotype, c.ref_reads, c.alt_reads] for c in cands_b]
        assert a == b
        assert max(x[2] for x in a) > 300


def test_alignment_index_rejects_unsorted_input(golden):
    g, _ = golden
    rows = list(reversed(g["rows"][:50]))
    bam = records.AlignmentFile(text=synth.genotype_sam_text(g["references"], g["lengths"], rows))
    with pytest.raises(ValueError):
        SVIM_genotyping.AlignmentIndex(bam)


def test_candidates_below_minimum_score_are_left_untouched(golden, oracle):
    g, bam = golden
    o = types.SimpleNamespace(**g["options"])
    low = _Candidate("DEL", "chr1", 1000, 2000, ["g00001"], o.minimum_score - 1)
    ok = _Candidate("DEL", "chr1", 1000, 2000, ["g00001"], o.minimum_score)
    SVIM_genotyping.genotype([low, ok], bam, "DEL", o, engine=oracle)
    assert (low.support_fraction, low.genotype, low.ref_reads, low.alt_reads) == (".", "./.", None, None)
    assert ok.alt_reads == 1 and ok.ref_reads is not None
