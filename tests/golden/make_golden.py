#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  The reference is pure Python; it is
imported from /root/reference/src with two third-party modules stubbed:

* ``pysam``  -> ``AlignedSegment`` = svim_amd.records.AlignedSegment (htslib coordinate rules),
               ``FastaFile`` = plain-text FASTA slicer with end clipping;
* ``edlib``  -> ``align(a, b)["editDistance"]`` = unit-cost global Levenshtein distance (unique value).

scipy (linkage / fcluster), random (MT19937 seed/sample) and statistics (mean/stdev) are the real
modules of this container (CPython 3.10.12, scipy 1.15.3); their versions are recorded in every file.

Outputs are DATA ONLY: inputs (SAM text, signature rows, condensed matrices) and the reference's
outputs for them.  No reference source is copied.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import gzip
import json
import os
import random
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, "/root/reference/src")

from svim_amd import records, synth   # noqa: E402


# ------------------------------------------------------------------ stubs
def levenshtein(a, b):
    """Unit-cost global edit distance (Myers/Hyyro bit-vector on Python ints)."""
    if len(a) < len(b):
        a, b = b, a
    m = len(b)
    if m == 0:
        return len(a)
    peq = {}
    for i, c in enumerate(b):
        peq[c] = peq.get(c, 0) | (1 << i)
    mask = (1 << m) - 1
    high = 1 << (m - 1)
    pv, mv, score = mask, 0, m
    for c in a:
        eq = peq.get(c, 0)
        xv = eq | mv
        xh = ((((eq & pv) + pv) ^ pv) | eq) & mask
        ph = (mv | ~(xh | pv)) & mask
        mh = pv & xh
        if ph & high:
            score += 1
        elif mh & high:
            score -= 1
        ph = ((ph << 1) | 1) & mask
        mh = (mh << 1) & mask
        pv = (mh | ~(xv | ph)) & mask
        mv = ph & xv
    return score


def levenshtein_dp(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb))
        prev = cur
    return prev[-1]


class FastaFile(object):
    def __init__(self, path):
        self.seqs = {}
        name = None
        opener = gzip.open if path.endswith(".gz") else open
        with opener(path, "rt") as fh:
            for line in fh:
                line = line.rstrip("\n")
                if line.startswith(">"):
                    name = line[1:].split()[0]
                    self.seqs[name] = []
                elif name is not None:
                    self.seqs[name].append(line)
        self.seqs = {k: "".join(v) for k, v in self.seqs.items()}

    def fetch(self, contig, start, end):
        return self.seqs[contig][start:end]

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


pysam_stub = types.ModuleType("pysam")
pysam_stub.AlignedSegment = records.AlignedSegment
pysam_stub.FastaFile = FastaFile
pysam_stub.AlignmentFile = records.AlignmentFile
sys.modules["pysam"] = pysam_stub
edlib_stub = types.ModuleType("edlib")
edlib_stub.align = lambda a, b, **kw: {"editDistance": levenshtein(a, b)}
sys.modules["edlib"] = edlib_stub

import numpy as np                                   # noqa: E402
import scipy                                         # noqa: E402
from scipy.cluster.hierarchy import linkage, fcluster   # noqa: E402
from svim import SVIM_intra, SVIM_inter, SVIM_COLLECT, SVIM_CLUSTER, SVIM_clustering, SVSignature  # noqa: E402

VERSIONS = {"python": sys.version.split()[0], "scipy": scipy.__version__, "numpy": np.__version__,
            "reference": "eldariont/svim v2.0.0 (/root/reference)"}


def options(**kw):
    o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10,
                              segment_overlap_tolerance=5, partition_max_distance=1000,
                              position_distance_normalizer=900, edit_distance_normalizer=1.0,
                              cluster_max_distance=0.5, all_bnds=False, genome=os.path.join(HERE, "ref.fa.gz"))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def opt_dict(o):
    d = dict(vars(o))
    d["genome"] = os.path.basename(d["genome"])
    return d


# ------------------------------------------------------------------ serialisation
def sig_row(s):
    t = s.type
    if t in ("DEL",):
        return [t, s.contig, s.start, s.end, s.signature, s.read]
    if t == "INS":
        return [t, s.contig, s.start, s.end, s.signature, s.read, s.sequence]
    if t == "INV":
        return [t, s.contig, s.start, s.end, s.signature, s.read, s.direction]
    if t == "DUP_TAN":
        return [t, s.contig, s.start, s.end, s.signature, s.read, s.copies, bool(s.fully_covered)]
    if t == "DUP_INT":
        return [t, s.contig1, s.start, s.end, s.signature, s.read, s.contig2, s.pos]
    if t == "BND":
        return [t, s.contig1, s.pos1, s.direction1, s.contig2, s.pos2, s.direction2, s.signature, s.read]
    raise ValueError(t)


def row_sig(r):
    t = r[0]
    S = SVSignature
    if t == "DEL":
        return S.SignatureDeletion(r[1], r[2], r[3], r[4], r[5])
    if t == "INS":
        return S.SignatureInsertion(r[1], r[2], r[3], r[4], r[5], r[6])
    if t == "INV":
        return S.SignatureInversion(r[1], r[2], r[3], r[4], r[5], r[6])
    if t == "DUP_TAN":
        return S.SignatureDuplicationTandem(r[1], r[2], r[3], r[6], r[7], r[4], r[5])
    if t == "DUP_INT":
        return S.SignatureInsertionFrom(r[1], r[2], r[3], r[6], r[7], r[4], r[5])
    if t == "BND":
        return S.SignatureTranslocation(r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8])
    raise ValueError(t)


def cluster_rows(clusters6, sigs):
    idx = {id(s): i for i, s in enumerate(sigs)}
    out = []
    for k, lst in enumerate(clusters6):
        rows = []
        for c in lst:
            members = [idx[id(m)] for m in c.members]
            if k < 3:
                rows.append([c.contig, c.start, c.end, c.score, c.size, c.std_span, c.std_pos, members])
            else:
                row = [c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end,
                       c.score, c.size, c.std_span, c.std_pos, members]
                if c.type == "BND":
                    row += [c.direction1, c.direction2]
                rows.append(row)
        out.append(rows)
    return out


def dump(name, obj):
    obj["versions"] = VERSIONS
    path = os.path.join(HERE, name)
    with gzip.GzipFile(path, "wb", mtime=0) if name.endswith(".gz") else open(path, "wb") as fh:
        fh.write(json.dumps(obj, separators=(",", ":")).encode("ascii"))
    print("wrote", name, os.path.getsize(path), "bytes")


# ------------------------------------------------------------------ G1: analyze_cigar_indel
def gen_intra():
    rng = random.Random(101)
    cases = []
    # the reference's own four known-answer vectors (src/tests/test_intra.py:9-22)
    kat = [
        ([(5, 10), (4, 20), (0, 10), (7, 10), (8, 5), (0, 5), (1, 50), (0, 30), (4, 25), (5, 15)], 30),
        ([(5, 10), (4, 20), (0, 30), (2, 50), (0, 30), (4, 25), (5, 15)], 30),
        ([(5, 10), (4, 20), (0, 30), (2, 40), (1, 50), (0, 30), (4, 25), (5, 15)], 30),
        ([(5, 10), (4, 20), (0, 30), (1, 40), (2, 50), (0, 30), (4, 25), (5, 15)], 30),
    ]
    for tuples, ml in kat:
        cases.append({"tuples": tuples, "min_length": ml,
                      "expect": [list(x) for x in SVIM_intra.analyze_cigar_indel(tuples, ml)]})
    for _ in range(1200):
        n = rng.choice((0, 1, 2, 5, 20, 64, 65, 200, 257, 700))
        ml = rng.choice((1, 30, 40, 41))
        tuples = []
        for _ in range(n):
            op = rng.choice((0, 0, 0, 1, 1, 2, 2, 3, 4, 5, 6, 7, 8, 9))
            ln = rng.choice((1, 2, 5, 29, 30, 39, 40, 41, 100, 5000, rng.randint(1, 60)))
            tuples.append((op, ln))
        cases.append({"tuples": tuples, "min_length": ml,
                      "expect": [list(x) for x in SVIM_intra.analyze_cigar_indel(tuples, ml)]})
    dump("g1_cigar_indel.json.gz", {"cases": cases,
                                    "source": "svim.SVIM_intra.analyze_cigar_indel (src/svim/SVIM_intra.py:8-30)"})


# ------------------------------------------------------------------ G2/G3: COLLECT
def run_collect(sam_text, opts, mode):
    bam = records.AlignmentFile(text=sam_text)
    fn = SVIM_COLLECT.analyze_alignment_file_coordsorted if mode == "coordinate" else \
        SVIM_COLLECT.analyze_alignment_file_querysorted
    sigs, bnds = fn(bam, opts)
    return sigs, bnds


def gen_collect(refs, references, lengths):
    # chimeric_read.sam is a data file of the reference's own test-suite (src/tests/chimeric_read.sam)
    with open("/root/reference/src/tests/chimeric_read.sam") as fh:
        chim = fh.read()
    with open(os.path.join(HERE, "chimeric_read.sam"), "w") as fh:
        fh.write(chim)
    bam = records.AlignmentFile(text=chim)
    alns = list(bam.fetch(until_eof=True))
    rebuilt = SVIM_COLLECT.retrieve_other_alignments(alns[0], bam)
    sa_table = [[a.cigarstring, a.reference_id, a.reference_start, a.reference_end, a.flag, a.mapping_quality,
                 a.query_alignment_start, a.query_alignment_end, a.infer_read_length()] for a in rebuilt]
    direct = [[a.cigarstring, a.reference_id, a.reference_start, a.reference_end, a.flag, a.mapping_quality,
               a.query_alignment_start, a.query_alignment_end, a.infer_read_length()] for a in alns[1:]]
    cases = []
    for all_bnds in (False, True):
        for mode in ("coordinate", "queryname"):
            o = options(all_bnds=all_bnds)
            sigs, bnds = run_collect(chim, o, mode)
            cases.append({"name": "chimeric", "sam_file": "chimeric_read.sam", "mode": mode, "options": opt_dict(o),
                          "signatures": [sig_row(s) for s in sigs], "bnds": [sig_row(s) for s in bnds]})
    dump("g3_satag.json", {"sa_rebuilt": sa_table, "supplementary_records": direct,
                           "source": "svim.SVIM_COLLECT.retrieve_other_alignments on src/tests/chimeric_read.sam "
                                     "(expected values of src/tests/test_satag.py:21-34)"})

    fuzz_sets = [
        ("fuzzA", dict(seed=11, n_reads=170, max_sv_size=100000), dict()),
        ("fuzzB", dict(seed=12, n_reads=170, max_sv_size=20000), dict(max_sv_size=20000, min_sv_size=30,
                                                                       segment_gap_tolerance=15,
                                                                       segment_overlap_tolerance=8)),
        ("fuzzC", dict(seed=13, n_reads=140, max_sv_size=100000), dict(min_mapq=5)),
        ("layoutD", dict(seed=14, n_reads=240, max_sv_size=20000), dict(max_sv_size=20000)),     # INV geometries + split-read insertions
    ]
    stats = {}
    for name, gen_kw, opt_kw in fuzz_sets:
        if name == "layoutD":
            recs = synth.inversion_insertion_layouts(references=references, lengths=lengths, **gen_kw)
        else:
            recs = synth.fuzz_split_reads(references=references, lengths=lengths, **gen_kw)
        for mode in ("coordinate", "queryname"):
            if mode == "coordinate":
                ordered = synth.coordinate_sort(recs)
            else:
                ordered = recs          # generator emits read by read: query-name grouped
            text = synth.sam_text(references, lengths, ordered, sort_order=mode)
            for all_bnds in (False, True):
                o = options(all_bnds=all_bnds, **opt_kw)
                sigs, bnds = run_collect(text, o, mode)
                for s in sigs:
                    stats[(s.type, s.signature)] = stats.get((s.type, s.signature), 0) + 1
                    if s.type == "INV":
                        stats[("INV", s.direction)] = stats.get(("INV", s.direction), 0) + 1
                cases.append({"name": name, "sam": text if not all_bnds else None, "mode": mode,
                              "options": opt_dict(o), "signatures": [sig_row(s) for s in sigs],
                              "bnds": [sig_row(s) for s in bnds]})
    # planted DEL/INS/INV reads: realistic C1-like slice
    recs = synth.planted_reads(21, 500, refs, references, lengths, n_sites=30, types=("DEL", "INS", "INV"))
    text = synth.sam_text(references, lengths, synth.coordinate_sort(recs))
    for all_bnds in (False, True):
        o = options(all_bnds=all_bnds)
        sigs, bnds = run_collect(text, o, "coordinate")
        for s in sigs:
            stats[(s.type, s.signature)] = stats.get((s.type, s.signature), 0) + 1
        cases.append({"name": "planted", "sam": text if not all_bnds else None, "mode": "coordinate",
                      "options": opt_dict(o), "signatures": [sig_row(s) for s in sigs],
                      "bnds": [sig_row(s) for s in bnds]})
    print("collect branch coverage:", sorted(stats.items()))
    dump("g2_collect.json.gz", {"cases": cases, "references": references, "lengths": lengths,
                                "source": "svim.SVIM_COLLECT.analyze_alignment_file_{coord,query}sorted "
                                          "(src/svim/SVIM_COLLECT.py:96-167) on synthetic SAM"})
    return cases


# ------------------------------------------------------------------ G4-G7: CLUSTER
def synth_signature_rows(seed, refs, references, lengths):
    """Direct signature lists that stress partition sizes (1,2,3,50,100,101,1045,1046,5000), same-read
    duplicates, exact ties, distances straddling 0.5, multi-contig string order, all six types."""
    rng = random.Random(seed)
    rows = []
    rid = [0]

    def read():
        rid[0] += 1
        return "r%d" % rid[0]

    def blob(typ, contig, center, n, span, jit_pos, jit_span, reads=None):
        for k in range(n):
            c = center + rng.randint(-jit_pos, jit_pos)
            sp = max(1, span + rng.randint(-jit_span, jit_span))
            st = max(0, c - sp // 2)
            rd = reads[k % len(reads)] if reads else read()
            src = rng.choice(("cigar", "suppl"))
            if typ == "DEL":
                rows.append(["DEL", contig, st, st + sp, src, rd])
            elif typ == "INS":
                base = blob.ins_seq.setdefault((contig, center), synth.random_seq(rng, span + jit_span + 5))
                seq = "".join(ch if rng.random() > 0.04 else rng.choice("ACGT") for ch in base[:sp])
                rows.append(["INS", contig, st, st + sp, src, rd, seq])
            elif typ == "INV":
                rows.append(["INV", contig, st, st + sp, "suppl", rd,
                             rng.choice(("left_fwd", "left_rev", "right_fwd", "right_rev"))])
            elif typ == "DUP_TAN":
                rows.append(["DUP_TAN", contig, st, st + sp, "suppl", rd, rng.randint(1, 4), rng.random() < 0.5])
            elif typ == "DUP_INT":
                c2 = rng.choice(references)
                rows.append(["DUP_INT", contig, st, st + sp, "suppl", rd, c2,
                             (center * 7) % 40000 + rng.randint(-jit_pos, jit_pos) + 500])
            elif typ == "BND":
                c2 = rng.choice(references)
                d1, d2 = rng.choice((("fwd", "fwd"), ("fwd", "rev"), ("rev", "rev"), ("rev", "fwd"), ("fwd", "fwd")))
                rows.append(["BND", contig, c, d1, c2, (center * 3) % 30000 + 200 + rng.randint(-jit_pos, jit_pos), d2,
                             "suppl", rd])
    blob.ins_seq = {}
    sizes = [1, 2, 3, 7, 50, 100, 101, 150]
    for typ in ("DEL", "INS", "INV", "DUP_TAN", "DUP_INT", "BND"):
        pos = 5000
        for contig in references:
            for n in sizes:
                if typ == "INS" and n > 60:
                    n = 60 if n == 100 else (101 if n == 101 else 40)
                span = rng.choice((60, 200, 800))
                if typ == "INS":
                    span = rng.choice((50, 120, 300))
                blob(typ, contig, pos, n, span, rng.choice((0, 5, 60, 300)), rng.choice((0, 3, 30)))
                # same-read duplicates + exact ties
                rd = read()
                blob(typ, contig, pos + 3, 3, span, 0, 0, reads=[rd])
                pos += rng.choice((900, 1500, 2500, 4000))
                if pos > min(lengths) - 3000:
                    pos = 5000 + rng.randint(0, 500)
    # very large DEL partitions: both random.sample code paths and RNG carry-over between partitions
    for n, center in ((1045, 20000), (1046, 40000), (3000, 60000)):
        blob("DEL", references[0], center, n, 300, 400, 100)
    rng.shuffle(rows)
    return rows


def gen_cluster(collect_cases, refs, references, lengths):
    cases = []
    # (i) cluster what COLLECT produced
    for c in collect_cases:
        if c["options"]["all_bnds"] or c["name"] == "chimeric":
            continue
        for kw in (dict(), dict(partition_max_distance=5000, cluster_max_distance=0.7,
                                position_distance_normalizer=450, edit_distance_normalizer=1.5)):
            if kw and c["mode"] != "coordinate":
                continue
            o = options(**{**{k: v for k, v in c["options"].items() if k != "genome"}, **kw})
            sigs = [row_sig(r) for r in c["signatures"]]
            res = SVIM_CLUSTER.cluster_sv_signatures(sigs, o)
            cases.append({"name": "from_collect:%s:%s" % (c["name"], c["mode"]), "options": opt_dict(o),
                          "signatures": c["signatures"], "clusters": cluster_rows(res, sigs)})
    # (ii) direct stress lists
    for seed, kw in ((31, dict()), (32, dict(partition_max_distance=300)), (33, dict(cluster_max_distance=0.3))):
        rows = synth_signature_rows(seed, refs, references, lengths)
        o = options(**kw)
        sigs = [row_sig(r) for r in rows]
        rows = [sig_row(s) for s in sigs]           # BND rows in canonical (post-constructor) form
        res = SVIM_CLUSTER.cluster_sv_signatures(sigs, o)
        cases.append({"name": "stress%d" % seed, "options": opt_dict(o), "signatures": rows,
                      "clusters": cluster_rows(res, sigs)})
        print("stress", seed, "n_sig", len(rows), "clusters", [len(x) for x in res])
    # form_partitions boundaries (G4)
    parts = []
    for case in cases[-3:]:
        sigs = [row_sig(r) for r in case["signatures"]]
        idx = {id(s): i for i, s in enumerate(sigs)}
        for typ in ("DEL", "INS", "INV", "DUP_TAN", "BND", "DUP_INT"):
            sub = [s for s in sigs if s.type == typ]
            p = SVIM_clustering.form_partitions(sub, case["options"]["partition_max_distance"])
            parts.append({"case": case["name"], "type": typ, "partitions": [[idx[id(s)] for s in q] for q in p]})
    dump("g4_partitions.json.gz", {"partitions": parts,
                                   "source": "svim.SVIM_clustering.form_partitions (src/svim/SVIM_clustering.py:17-29)"})
    dump("g5_cluster.json.gz", {"cases": cases, "references": references, "lengths": lengths,
                                "source": "svim.SVIM_CLUSTER.cluster_sv_signatures (src/svim/SVIM_CLUSTER.py:7-26)"})
    # G6: span_position_distance bit patterns
    import struct
    pairs = []
    ref = FastaFile(os.path.join(HERE, "ref.fa.gz"))
    rows = cases[-3]["signatures"]
    sigs = [row_sig(r) for r in rows]
    rng = random.Random(5)
    by_type = {}
    for i, s in enumerate(sigs):
        by_type.setdefault(s.type, []).append(i)
    for typ, ids in by_type.items():
        for _ in range(150):
            i, j = rng.sample(ids, 2)
            if typ == "INS" and (abs(sigs[i].start - sigs[j].start) > 3000 or sigs[i].contig != sigs[j].contig):
                continue
            d = SVIM_clustering.span_position_distance(sigs[i], sigs[j], typ, ref, 900, 1.0, 0.5)
            pairs.append([i, j, struct.pack("<d", float(d)).hex()])
    dump("g6_distance.json.gz", {"signatures": rows, "pairs": pairs, "params": [900, 1.0, 0.5],
                                 "source": "svim.SVIM_clustering.span_position_distance (src/svim/SVIM_clustering.py:47-96)"})


# ------------------------------------------------------------------ linkage / rng / edit distance
def gen_linkage():
    rng = random.Random(77)
    cases = []
    for it in range(240):
        n = rng.choice((2, 3, 4, 5, 8, 13, 30, 64, 65, 100))
        pts = [(rng.randint(0, 40) * rng.choice((1, 1, 25)), rng.choice((50, 60, 100, 400))) for _ in range(n)]
        if it % 3 == 0:
            pts = [pts[rng.randrange(max(1, n // 3))] for _ in range(n)]       # many exact duplicates
        reads = [rng.randrange(max(2, n // 2)) for _ in range(n)]
        d = []
        for i in range(n - 1):
            for j in range(i + 1, n):
                if it % 2 and reads[i] == reads[j]:
                    d.append(99999)
                else:
                    (c1, s1), (c2, s2) = pts[i], pts[j]
                    d.append(abs(c1 - c2) / 900 + abs(s1 - s2) / max(s1, s2))
        Z = linkage(np.array(d), method="average")
        t = rng.choice((0.5, 0.5, 0.3, 0.0, 1.0))
        lab = [int(x) for x in fcluster(Z, t, criterion="distance")]
        cases.append({"n": n, "d": [float(x).hex() for x in d], "t": t,
                      "Z": [[float(v).hex() for v in row] for row in Z], "labels": lab})
    dump("g_linkage.json.gz", {"cases": cases,
                               "source": "scipy.cluster.hierarchy.linkage(method='average') + fcluster(criterion="
                                         "'distance') as called at src/svim/SVIM_clustering.py:170-171"})


def gen_rng():
    out = {"getrandbits": {}, "samples": []}
    for k in (7, 8, 10, 11, 13, 32):
        random.seed(1524)
        out["getrandbits"][str(k)] = [random.getrandbits(k) for _ in range(2000)]
    random.seed(1524)
    for n in (101, 150, 1045, 1046, 5000, 77777, 128, 100000, 101):
        out["samples"].append({"n": n, "idx": random.sample(range(n), 100)})
    dump("g8_rng.json.gz", {"rng": out, "source": "random.seed(1524); random.sample(partition, 100) as at "
                                                  "src/svim/SVIM_clustering.py:129,133 (sequential, one seed)"})


def gen_edit():
    rng = random.Random(9)
    cases = []
    for _ in range(300):
        la = rng.choice((0, 1, 5, 31, 32, 33, 63, 64, 65, 100, 200, 400))
        a = synth.random_seq(rng, la)
        mode = rng.random()
        if mode < 0.5:
            b = list(a)
            for _ in range(rng.randint(0, max(1, la // 6))):
                r = rng.random()
                p = rng.randrange(len(b) + 1)
                if r < 0.33 and b:
                    b[min(p, len(b) - 1)] = rng.choice("ACGTN")
                elif r < 0.66 and b:
                    del b[min(p, len(b) - 1)]
                else:
                    b.insert(p, rng.choice("ACGT"))
            b = "".join(b)
        else:
            b = synth.random_seq(rng, rng.choice((0, 3, 64, 130, 333)))
        d = levenshtein_dp(a, b)
        assert d == levenshtein(a, b) == levenshtein(b, a)
        cases.append([a, b, d])
    dump("g_editdistance.json.gz", {"cases": cases,
                                    "source": "textbook O(nm) Levenshtein DP: the value edlib.align(a,b)['editDistance'] "
                                              "returns at src/svim/SVIM_clustering.py:45 (edlib not installed; unique by mathematics)"})


def gen_c1():
    """BASELINE.json configs[0]: 10k-read synthetic 1-contig input, DEL/INS only, through the reference's CPU path
    (analyze_alignment_file_coordsorted + cluster_sv_signatures), timed.  The reads are regenerated from the seed by the
    tests (svim_amd.synth is deterministic); only the reference's outputs are stored."""
    import time
    contigs = [("chr1", 2000000)]
    refs = synth.make_reference(7, contigs)
    recs = synth.coordinate_sort(synth.planted_reads(8, 10000, refs, ["chr1"], [2000000], n_sites=300, types=("DEL", "INS"),
                                                     read_len=(1000, 6000)))
    fa = os.path.join(HERE, "_c1.fa")
    synth.write_fasta(fa, refs, lower_every=0)
    text = synth.sam_text(["chr1"], [2000000], recs)
    o = options(genome=fa)
    bam = records.AlignmentFile(text=text)
    t0 = time.perf_counter()
    sigs, bnds = SVIM_COLLECT.analyze_alignment_file_coordsorted(bam, o)
    t1 = time.perf_counter()
    res = SVIM_CLUSTER.cluster_sv_signatures(sigs, o)
    t2 = time.perf_counter()
    os.remove(fa)
    n_ops = sum(len(a.cigartuples) for a in recs)
    print("C1: %d records, %d ops, %d signatures, collect %.2fs cluster %.2fs" % (len(recs), n_ops, len(sigs), t1 - t0, t2 - t1))
    dump("g_c1.json.gz", {"note": "BASELINE.json configs[0] (10k-read synthetic 1-contig BAM, DEL/INS only).  Deviation from SURVEY.md section 8(d) C1: a 2 Mb "
                                  "contig, reads of 1-6 kb and 4.1 M CIGAR operations instead of 250 Mb / up to 20 kb / ~2*10^7 operations - the reference's "
                                  "Python loops need 7 s for this size (reference_seconds), which keeps regenerating the golden practical; same coverage "
                                  "per site, same planted-site density per read",
                          "generator": "synth.planted_reads(8, 10000, make_reference(7, [('chr1', 2000000)]), ['chr1'], [2000000], n_sites=300, "
                                       "types=('DEL','INS'), read_len=(1000,6000)) coordinate-sorted",
                          "n_records": len(recs), "n_ops": n_ops, "options": opt_dict(options()),
                          "signatures": [sig_row(s) for s in sigs], "clusters": cluster_rows(res, sigs),
                          "reference_seconds": {"collect": t1 - t0, "cluster": t2 - t1,
                                                "note": "reference Python functions in the build container, 1 core, pysam/edlib stubbed "
                                                        "(edlib = pure-Python bit-vector Levenshtein, so cluster time is an upper bound)"},
                          "source": "svim.SVIM_COLLECT.analyze_alignment_file_coordsorted + svim.SVIM_CLUSTER.cluster_sv_signatures"})


def records_from_hostbatch(hb, name_fmt="r%08d"):
    """HostBatch (numpy SoA) -> record objects with pysam's attribute names (primaries without SA tags: what C1 holds)"""
    A = hb.arrays
    lut = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
    out = []
    co, so = A["cigar_off"].astype(np.int64), A["seq_off"].astype(np.int64)
    for i in range(hb.n_rec):
        a = records.AlignedSegment()
        a.query_name = name_fmt % int(A["read_id"][i])
        a.flag = int(A["flag"][i]) & 0x0fff
        a.reference_id, a.reference_start, a.mapping_quality = int(A["tid"][i]), int(A["pos"][i]), int(A["mapq"][i])
        w = A["cigar"][co[i]:co[i + 1]].astype(np.int64)
        a.cigartuples = list(zip((w & 15).tolist(), (w >> 4).tolist()))
        sb = A["seq"][so[i]:so[i + 1]]
        nib = np.empty(2 * sb.shape[0], dtype=np.uint8)
        nib[0::2], nib[1::2] = sb >> 4, sb & 15
        a.query_sequence = lut[nib[:int(A["lseq"][i])]].tobytes().decode("ascii")
        out.append(a)
    return out


def gen_c1_full():
    """BASELINE.json configs[0] AT ITS STATED SIZE: 250 Mb contig, 10 000 reads of length
    ~ triangular(100, 20000, 15000), ~1.2*10^7 CIGAR operations, 300 planted DEL/INS sites - through the reference's CPU path, timed.  The
    input is regenerated from its seed by the tests (tests/helpers.py:c1_full_case); only the reference's outputs are stored."""
    import time
    import helpers as H

    class BamStub(object):                                 # the duck type analyze_alignment_file_coordsorted needs (src/svim/SVIM_COLLECT.py:132-167)
        def __init__(self, recs):
            self.recs, self.references = recs, ["chr1"]

        def fetch(self, until_eof=True):
            return iter(self.recs)

        def getrname(self, tid):
            return self.references[tid]

        get_reference_name = getrname

        def get_tid(self, name):
            return self.references.index(name)
    hb, genome, meta = H.c1_full_case()
    recs = records_from_hostbatch(hb)
    fa = os.path.join(HERE, "_c1_full.fa")
    H.write_fasta_from_codes(fa, "chr1", genome)
    o = options(genome=fa)
    t0 = time.perf_counter()
    sigs, bnds = SVIM_COLLECT.analyze_alignment_file_coordsorted(BamStub(recs), o)
    t1 = time.perf_counter()
    res = SVIM_CLUSTER.cluster_sv_signatures(sigs, o)
    t2 = time.perf_counter()
    os.remove(fa)
    print("C1 full: %d records, %d ops, %d signatures, %d clusters, collect %.2fs cluster %.2fs" % (len(recs), meta["n_ops"], len(sigs), sum(len(x) for x in res), t1 - t0, t2 - t1))
    dump("g_c1_full.json.gz", {"note": "BASELINE.json configs[0] at the size SURVEY.md section 8(d) C1 states (250 Mb contig, reads ~ triangular(100, 20000, 15000), "
                                       "300 planted DEL/INS sites of 50-5000 bp)",
                               "generator": "tests/helpers.py:c1_full_case = svim_amd.devsynth.make_batch(device='cpu', **%r)" % (H.C1_FULL,),
                               "n_records": len(recs), "n_ops": int(meta["n_ops"]), "options": opt_dict(options()),
                               "signatures": [sig_row(s) for s in sigs], "n_bnds": len(bnds), "clusters": cluster_rows(res, sigs),
                               "reference_seconds": {"collect": t1 - t0, "cluster": t2 - t1, "records_per_s": len(recs) / (t2 - t0),
                                                     "note": "the reference's Python functions in the build container, 1 core, pysam/edlib stubbed (edlib = pure-Python "
                                                             "bit-vector Levenshtein: cluster time is an upper bound)"},
                               "source": "svim.SVIM_COLLECT.analyze_alignment_file_coordsorted + svim.SVIM_CLUSTER.cluster_sv_signatures"})


def gen_c1_bench_sample():
    """The reference ITSELF on a configs[1]-SHAPED sample (VERDICT r04 item 9 / SURVEY.md section 8d-i): 20 000 ONT-like reads (N50 20 kb) on a 5 Mb contig
    with configs[1]'s densities - DEL / INS / INV sites, 12 % of the reads split across an inversion (supplementary records + SA tags) - written as a BAM
    file (svim_amd.harness.write_bam_from_batch), read back as record objects and handed to the reference's COLLECT and CLUSTER, timed.  Stored: the
    reference's signatures and clusters (a parity fixture at the bench workload's shape) and its seconds (bench.py's cpu_baseline.reference_python)."""
    import time
    import helpers as H
    from svim_amd import harness
    hb, genome, meta = H.c1_bench_sample_case()
    bam = os.path.join(HERE, "_c1_bench_sample.bam")
    harness.write_bam_from_batch(bam, hb, ["chr1"], [int(genome.shape[0])])
    af = records.AlignmentFile(bam)
    recs = list(af.fetch(until_eof=True))

    class BamStub(object):
        def __init__(self, recs):
            self.recs, self.references = recs, ["chr1"]

        def fetch(self, until_eof=True):
            return iter(self.recs)

        def getrname(self, tid):
            return self.references[tid]

        get_reference_name = getrname

        def get_tid(self, name):
            return self.references.index(name) if name in self.references else -1
    fa = os.path.join(HERE, "_c1_bench_sample.fa")
    H.write_fasta_from_codes(fa, "chr1", genome)
    o = options(genome=fa)
    t0 = time.perf_counter()
    sigs, bnds = SVIM_COLLECT.analyze_alignment_file_coordsorted(BamStub(recs), o)
    t1 = time.perf_counter()
    print("C1 bench sample: COLLECT %.1f s, %d signatures" % (t1 - t0, len(sigs)), flush=True)
    res = SVIM_CLUSTER.cluster_sv_signatures(sigs, o)
    t2 = time.perf_counter()
    os.remove(fa)
    os.remove(bam)
    n_reads = sum(1 for a in recs if not (a.flag & (256 | 2048)))
    print("C1 bench sample: %d records (%d reads), %d ops, %d signatures, %d clusters, collect %.2fs cluster %.2fs" % (
        len(recs), n_reads, meta["n_ops"], len(sigs), sum(len(x) for x in res), t1 - t0, t2 - t1))
    dump("g_c1_bench_sample.json.gz", {
        "note": "BASELINE.json configs[1] at a fiftieth of its size and the same densities (tests/helpers.py:c1_bench_sample_case) through the reference",
        "generator": "tests/helpers.py:c1_bench_sample_case = svim_amd.devsynth.make_batch(device='cpu', **%r)" % (H.C1_BENCH_SAMPLE,),
        "n_records": len(recs), "n_reads": n_reads, "n_ops": int(meta["n_ops"]), "options": opt_dict(options()),
        "signatures": [sig_row(s) for s in sigs], "n_bnds": len(bnds), "clusters": cluster_rows(res, sigs),
        "reference_seconds": {"collect": t1 - t0, "cluster": t2 - t1, "records_per_s": len(recs) / (t2 - t0), "reads_per_s": n_reads / (t2 - t0),
                              "note": "the reference's Python functions in the build container, 1 core, pysam/edlib stubbed (records parsed before the clock starts; "
                                      "edlib = pure-Python bit-vector Levenshtein: the cluster time is an upper bound)"},
        "source": "svim.SVIM_COLLECT.analyze_alignment_file_coordsorted + svim.SVIM_CLUSTER.cluster_sv_signatures"})


def gen_entrypoints(collect_cases):
    """Per-read entry points (analyze_alignment_indel, analyze_read_segments) and the COMBINE-side re-clustering
    (partition_and_cluster_candidates) as the reference computes them."""
    from svim import SVCandidate
    case = [c for c in collect_cases if c["name"] == "fuzzA" and c["mode"] == "coordinate" and c.get("sam")][0]
    bam = records.AlignmentFile(text=case["sam"])
    recs = list(bam.fetch(until_eof=True))
    out = []
    for all_bnds in (False, True):
        o = options(all_bnds=all_bnds)
        per = []
        for i, a in enumerate(recs[:400]):
            if a.is_unmapped or a.is_secondary:
                continue
            s1, b1 = SVIM_intra.analyze_alignment_indel(a, bam, a.query_name, o)
            entry = {"rec": i, "indel": [sig_row(s) for s in s1], "indel_bnd": [sig_row(s) for s in b1]}
            if not a.is_supplementary:
                sup = [x for x in SVIM_COLLECT.retrieve_other_alignments(a, bam) if x.mapping_quality >= o.min_mapq]
                s2, b2 = SVIM_inter.analyze_read_segments(a, sup, bam, o)
                entry["segments"] = [sig_row(s) for s in s2]
                entry["segments_bnd"] = [sig_row(s) for s in b2]
            per.append(entry)
        out.append({"all_bnds": all_bnds, "options": opt_dict(o), "per_record": per})
    # candidates: DUP_INT candidates scattered so that partitions of 1, 2, several and > 100 members occur
    rng = random.Random(17)
    cand_rows = []
    pos = 20000
    for n in (1, 2, 5, 30, 130, 3, 1, 101):
        for k in range(n):
            s = pos + rng.randint(-60, 60)
            ln = rng.choice((300, 320, 900))
            d = 70000 + (pos // 7) % 30000 + rng.randint(-80, 80)
            cand_rows.append(["chr1", s, s + ln, "chr2", d, d + ln, rng.randint(1, 40), rng.choice((None, 3.5, 10.25)), rng.choice((None, 1.5, 7.0)),
                              rng.random() < 0.2, ["m%d_%d" % (pos, k)]])
        pos += rng.choice((700, 2500, 6000))
    rng.shuffle(cand_rows)
    cands = [SVCandidate.CandidateDuplicationInterspersed(r[0], r[1], r[2], r[3], r[4], r[5], list(r[10]), r[6], r[7], r[8], r[9]) for r in cand_rows]
    o = options()
    res = SVIM_clustering.partition_and_cluster_candidates(cands, o, "interspersed duplication candidates")
    res_rows = [[c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end, c.score, c.std_span, c.std_pos,
                 bool(c.cutpaste), list(c.members)] for c in res]
    dump("g_entrypoints.json.gz", {"sam_case": "fuzzA/coordinate of g2_collect.json.gz", "runs": out, "candidates": cand_rows,
                                   "merged_candidates": res_rows, "options": opt_dict(o),
                                   "source": "svim.SVIM_intra.analyze_alignment_indel, svim.SVIM_inter.analyze_read_segments, "
                                             "svim.SVIM_clustering.partition_and_cluster_candidates"})


def gen_genotype():
    """GENOTYPE: the reference's genotype() on synthetic alignments; the AlignmentFile stand-in answers
    fetch(contig, start, stop) with htslib's overlap rule (svim_amd/records.py)."""
    from svim import SVIM_genotyping, SVCandidate
    rng = random.Random(77)
    references, lengths = ["chr1", "chr2"], [60000, 30000]
    rows = synth.genotype_rows(5, lengths)
    bam = records.AlignmentFile(text=synth.genotype_sam_text(references, lengths, rows))
    o = options(minimum_score=3, minimum_depth=4, homozygous_threshold=0.8, heterozygous_threshold=0.2)
    by_tid = {t: [r for r in rows if r[2] == t] for t in range(len(lengths))}
    cases = []
    for typ in ("DEL", "INV", "INS", "DUP_INT"):
        cand_rows = []
        for k in range(70):
            tid = rng.randrange(2)
            span = rng.choice((45, 120, 900, 3000, 4100, 7000))
            if k % 9 == 0:
                start = 20000 - rng.randint(0, 400) if tid == 0 else rng.randint(0, 800)         # the deep pile / the contig start
            elif k % 9 == 1:
                start = lengths[tid] - span - rng.randint(1, 600)                                # the contig end
            else:
                start = rng.randint(0, lengths[tid] - span - 1)
            near = [r for r in by_tid[tid] if r[3] < start + span + 300 and r[3] + r[5] > start - 300 and not (r[1] & 4)]
            rng.shuffle(near)
            members = [r[0] for r in near[:rng.choice((0, 1, 2, 3, 6, 12, 40))]]
            if members and rng.random() < 0.3:
                members.append(members[0])                                                       # two signatures of one read
            score = rng.choice((1, 2, 3, 4, 10, 40))
            cand_rows.append([references[tid], start, start + span, members, score])
        cands = []
        for contig, s, e, members, score in cand_rows:
            sigs = [SVSignature.SignatureDeletion(contig, s, e, "cigar", m) for m in members]
            if typ == "DEL":
                cands.append(SVCandidate.CandidateDeletion(contig, s, e, sigs, score, 1.0, 1.0))
            elif typ == "INV":
                cands.append(SVCandidate.CandidateInversion(contig, s, e, sigs, score, 1.0, 1.0))
            elif typ == "INS":
                cands.append(SVCandidate.CandidateNovelInsertion(contig, s, e, "", sigs, score, 1.0, 1.0))
            else:
                cands.append(SVCandidate.CandidateDuplicationInterspersed("chr1", 100, 100 + e - s, contig, s, e, sigs, score, 1.0, 1.0))
        SVIM_genotyping.genotype(cands, bam, typ, o)
        cases.append({"type": typ, "candidates": cand_rows,
                      "expected": [[c.support_fraction, c.genotype, c.ref_reads, c.alt_reads] for c in cands]})
    dump("g_genotype.json.gz", {"references": references, "lengths": lengths, "rows": rows, "options": opt_dict(o), "cases": cases,
                                "source": "svim.SVIM_genotyping.genotype (reads via svim_amd.records.AlignmentFile.fetch: htslib overlap rule)"})


def _read_tree(d):
    out = {}
    for root, _, files in os.walk(d):
        for f in sorted(files):
            with open(os.path.join(root, f)) as fh:
                out[os.path.relpath(os.path.join(root, f), d)] = fh.read()
    return out


def gen_writers():
    """The reference's own writers (src/svim/SVIM_CLUSTER.py:29-106) on the reference's clusters of g5's 'stress31' case - and, as a
    check at generation time, the SAME reference writers driven by OUR objects (CPU path: oracle tables -> svim_amd objects): the text
    must agree except for the last digits of the FP columns (statistics.stdev vs the FP64 two-pass formula)."""
    import tempfile
    from svim_amd import _abi, batch as sbatch, convert
    from oracle import oracle as om
    import helpers as H
    g5 = H.load("g5_cluster.json.gz")
    case = [c for c in g5["cases"] if c["name"] == "stress31"][0]
    o = options(**{k: v for k, v in case["options"].items() if k != "genome"})
    sigs = [row_sig(r) for r in case["signatures"]]
    res = SVIM_CLUSTER.cluster_sv_signatures(sigs, o)
    with tempfile.TemporaryDirectory() as d:
        SVIM_CLUSTER.write_signature_clusters_bed(d, res)
        SVIM_CLUSTER.write_signature_clusters_vcf(d, res, "2.0.0")
        ref_text = _read_tree(d)
    ours_sigs = [H.row_sig(r) for r in case["signatures"]]
    tab, contigs, reads = convert.sigtable_from_objects(ours_sigs, convert.Interner(g5["references"]))
    orc = om.Oracle()
    off, codes = convert.genome_arrays(os.path.join(HERE, "ref.fa.gz"), contigs.names)
    orc.set_genome(off, codes)
    ct = orc.cluster(_abi.Params.from_options(o), sbatch.contig_ranks(contigs.names), table=tab)
    ours = convert.cluster_objects(ct, ours_sigs, contigs.names)
    with tempfile.TemporaryDirectory() as d:
        SVIM_CLUSTER.write_signature_clusters_bed(d, ours)            # the REFERENCE's writers, our objects
        SVIM_CLUSTER.write_signature_clusters_vcf(d, ours, "2.0.0")
        our_text = _read_tree(d)
    assert sorted(ref_text) == sorted(our_text)
    for name in ref_text:
        assert H.text_close(our_text[name], ref_text[name]) is None, (name, H.text_close(our_text[name], ref_text[name]))
    dump("g_writers.json.gz", {"case": "stress31 of g5_cluster.json.gz", "version": "2.0.0", "files": ref_text,
                               "source": "svim.SVIM_CLUSTER.write_signature_clusters_bed / _vcf (src/svim/SVIM_CLUSTER.py:29-106) on the reference's "
                                         "clusters; the same writers driven by svim_amd's objects produced the same text at generation time"})


# ------------------------------------------------------------------ COMBINE-side consumers of the cluster lists
def cand_row(c, idx):
    """Candidate object (src/svim/SVCandidate.py) -> plain row: every data attribute by name, members as signature indices."""
    d = {k: v for k, v in vars(c).items() if k not in ("members", "complement")}
    d["members"] = [idx[id(m)] for m in c.members]
    d["class"] = type(c).__name__
    return d


def combine_signature_rows(seed):
    """A signature list whose clusters drive every branch of merge_translocations_at_insertions / flag_cutpaste_candidates /
    combine_clusters: insertions flanked by fwd-fwd + rev-rev breakends whose destinations are one insertion length apart (both
    for an insertion on the canonically first contig and on the second one, where only the REVERSED breakend clusters match), a
    flanked insertion with the wrong destination distance, an insertion on a contig without rev-rev breakends (the KeyError path),
    interspersed duplications with and without a deletion of their source, insertions that coincide with an interspersed /
    a tandem duplication, plus ordinary deletions, inversions and breakends."""
    rng = random.Random(seed)
    rows, rid = [], [0]

    def read():
        rid[0] += 1
        return "q%d" % rid[0]

    def j(x, k=3):
        return x + rng.randint(-k, k)

    def ins(contig, pos, length, n, reads=None):
        base = synth.random_seq(rng, length + 8)
        out = []
        for k in range(n):
            rd = reads[k] if reads else read()
            s = j(pos)
            ln = max(40, j(length, 2))
            seq = "".join(ch if rng.random() > 0.03 else rng.choice("ACGT") for ch in base[:ln])
            rows.append(["INS", contig, s, s + ln, "cigar", rd, seq])
            out.append(rd)
        return out

    def bnd(c1, p1, d1, c2, p2, d2, n, reads=None):
        for k in range(n):
            rows.append(["BND", c1, j(p1), d1, c2, j(p2), d2, "suppl", reads[k] if reads else read()])

    # A: chr1:30000, 300 bp inserted, copied from chr2:10000-10300
    bnd("chr1", 30000, "fwd", "chr2", 10000, "fwd", 6)
    bnd("chr2", 10300, "fwd", "chr1", 30000, "fwd", 6)
    ins("chr1", 30000, 300, 8)
    # B: chr2:40000, 200 bp inserted, copied from chr1:90000-90200 (the matching breakend clusters exist only reversed)
    bnd("chr2", 40000, "fwd", "chr1", 90000, "fwd", 5)
    bnd("chr1", 90200, "fwd", "chr2", 40000, "fwd", 5)
    ins("chr2", 40000, 200, 7)
    # C: flanked, but the destinations are 900 bp apart for a 250 bp insertion
    bnd("chr1", 60000, "fwd", "chr2", 25000, "fwd", 5)
    bnd("chr2", 25900, "fwd", "chr1", 60000, "fwd", 5)
    ins("chr1", 60000, 250, 6)
    # D: chr10 carries fwd-fwd breakends only
    bnd("chr10", 20000, "fwd", "chr2", 50000, "fwd", 5)
    ins("chr10", 20000, 150, 6)
    # E: breakends too far from the insertion (> trans_sv_max_distance)
    bnd("chr1", 120000, "fwd", "chr2", 30000, "fwd", 4)
    bnd("chr2", 30180, "fwd", "chr1", 120000, "fwd", 4)
    ins("chr1", 121000, 180, 6)
    # interspersed duplications from split reads: F with its source deleted (cut & paste), G without
    for k in range(6):
        s = j(50000)
        rows.append(["DUP_INT", "chr1", s, s + j(500, 2), "suppl", read(), "chr2", j(20000)])
    for k in range(6):
        s = j(50000)
        rows.append(["DEL", "chr1", s, s + j(500, 2), "cigar", read()])
    for k in range(5):
        s = j(140000)
        rows.append(["DUP_INT", "chr1", s, s + j(700, 2), "suppl", read(), "chr10", j(30000)])
    # an insertion where duplication F was pasted (same length): removed from the insertion list
    ins("chr2", 20000, 500, 6)
    # a tandem duplication and an insertion that is really that duplication
    for k in range(6):
        s = j(100000)
        rows.append(["DUP_TAN", "chr1", s, s + j(400, 2), "suppl", read(), 1, rng.random() < 0.5])
    ins("chr1", 100400, 400, 5)
    # ordinary calls
    for k in range(7):
        s = j(15000)
        rows.append(["DEL", "chr2", s, s + j(800, 3), rng.choice(("cigar", "suppl")), read()])
    for k in range(6):
        s = j(150000)
        rows.append(["INV", "chr1", s, s + j(2000, 3), "suppl", read(), ("left_fwd", "right_fwd", "left_rev", "right_rev")[k % 4]])
    ins("chr10", 45000, 90, 9)
    bnd("chr1", 170000, "fwd", "chr10", 55000, "rev", 5)
    bnd("chr2", 5000, "rev", "chr10", 5000, "fwd", 3)
    rng.shuffle(rows)
    return rows


def gen_combine():
    """The reference's COMBINE-side consumers of the six cluster lists - merge_translocations_at_insertions and
    flag_cutpaste_candidates (src/svim/SVIM_merging.py:93-159, :12-29) and the whole combine_clusters (src/svim/SVIM_COMBINE.py:332-478,
    --skip_consensus: no spoa) - on the reference's own clusters; and, as a check at generation time, THE SAME reference functions
    driven by svim_amd's lazy ClusterLists (CPU path: oracle tables), which must give the same candidates although the functions
    delete from / extend the lists they are handed."""
    for name in ("spoa", "cpuinfo"):
        stub = types.ModuleType(name)
        stub.poa = stub.get_cpu_info = None                     # never called with skip_consensus
        sys.modules.setdefault(name, stub)
    from svim import SVIM_COMBINE, SVIM_merging
    from svim_amd import _abi, batch as sbatch, convert
    from oracle import oracle as om
    import helpers as H
    references = ["chr1", "chr2", "chr10"]
    rows = combine_signature_rows(4711)
    o = options(trans_sv_max_distance=500, del_ins_dup_max_distance=1.0, skip_consensus=True)
    sigs = [row_sig(r) for r in rows]
    idx = {id(s): i for i, s in enumerate(sigs)}
    res = SVIM_CLUSTER.cluster_sv_signatures(sigs, o)
    clusters = cluster_rows(res, sigs)

    def run(c6, idx):
        dele, insr, inv, tan, dint, bnd = [list(x) if isinstance(x, list) else x for x in c6]
        if isinstance(bnd, list):
            b2, i2 = list(bnd), list(insr)
        else:
            import copy
            b2, i2 = copy.copy(bnd), copy.copy(insr)
        new_from, to_remove = SVIM_merging.merge_translocations_at_insertions(b2, i2, o)
        merged = [[c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end, c.score, c.size,
                   [idx[id(m)] for m in c.members], c.type, c.std_span, c.std_pos] for c in new_from]
        flagged = [cand_row(c, idx) for c in SVIM_merging.flag_cutpaste_candidates(list(dint) + new_from, dele, o)]
        n_ins_before = len(insr)
        out = SVIM_COMBINE.combine_clusters((dele, insr, inv, tan, dint, bnd), o)
        return {"merged_insertion_from_clusters": merged, "inserted_regions_to_remove": to_remove, "n_bnd_after_merge": len(b2),
                "flag_cutpaste": flagged, "n_ins_before": n_ins_before, "n_ins_after": len(insr), "n_dup_int_after": len(dint),
                "combine": [[cand_row(c, idx) for c in lst] for lst in out]}
    exp = run(res, idx)
    assert len(exp["merged_insertion_from_clusters"]) >= 2 and exp["inserted_regions_to_remove"], exp["merged_insertion_from_clusters"]
    assert {r["cutpaste"] for r in exp["flag_cutpaste"]} == {True, False}
    assert exp["n_ins_before"] - exp["n_ins_after"] > len(exp["inserted_regions_to_remove"])          # the overlap rules removed some too
    # the same reference code on OUR lists
    ours_sigs = [H.row_sig(r) for r in rows]
    oidx = {id(s): i for i, s in enumerate(ours_sigs)}
    tab, contigs, reads = convert.sigtable_from_objects(ours_sigs, convert.Interner(references))
    orc = om.Oracle()
    off, codes = convert.genome_arrays(os.path.join(HERE, "ref.fa.gz"), contigs.names)
    orc.set_genome(off, codes)
    ct = orc.cluster(_abi.Params.from_options(o), sbatch.contig_ranks(contigs.names), table=tab)
    got = run(convert.cluster_objects(ct, ours_sigs, contigs.names), oidx)
    diff = H.first_json_difference(got, exp)
    assert diff is None, diff
    dump("g_combine.json.gz", {"references": references, "signatures": rows, "options": opt_dict(o), "clusters": clusters, "expected": exp,
                               "source": "svim.SVIM_merging.merge_translocations_at_insertions / flag_cutpaste_candidates (src/svim/SVIM_merging.py:93-159, "
                                         ":12-29) and svim.SVIM_COMBINE.combine_clusters (src/svim/SVIM_COMBINE.py:332-478, skip_consensus) on the "
                                         "reference's clusters; the same functions driven by svim_amd's ClusterLists gave the same rows at generation time"})


def main():
    contigs = [("chr1", 180000), ("chr2", 60000), ("chr10", 60000)]   # tid order != Python string order
    refs = synth.make_reference(1, contigs)
    references = [c[0] for c in contigs]
    lengths = [c[1] for c in contigs]
    fa = os.path.join(HERE, "ref.fa")
    synth.write_fasta(fa, refs)
    with open(fa, "rb") as f_in, gzip.GzipFile(fa + ".gz", "wb", mtime=0) as f_out:
        f_out.write(f_in.read())
    os.remove(fa)
    gen_intra()
    collect_cases = gen_collect(refs, references, lengths)
    gen_cluster(collect_cases, refs, references, lengths)
    gen_entrypoints(collect_cases)
    gen_linkage()
    gen_rng()
    gen_edit()
    gen_c1()
    gen_c1_full()
    gen_c1_bench_sample()
    gen_genotype()
    gen_writers()
    gen_combine()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "genotype":
        gen_genotype()                 # this fixture only (the others are untouched)
    elif len(sys.argv) > 1 and sys.argv[1] == "writers":
        gen_writers()
    elif len(sys.argv) > 1 and sys.argv[1] == "c1_full":
        gen_c1_full()
    elif len(sys.argv) > 1 and sys.argv[1] == "combine":
        gen_combine()
    elif len(sys.argv) > 1 and sys.argv[1] == "c1_bench_sample":
        gen_c1_bench_sample()
    else:
        main()
