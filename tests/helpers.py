"""Shared test helpers: golden-vector loading and row <-> table conversion."""
import gzip
import json
import os
import types

import numpy as np

from svim_amd import _abi, convert, records, batch
from svim_amd.signatures import (SignatureDeletion, SignatureInsertion, SignatureInversion, SignatureInsertionFrom,
                                 SignatureDuplicationTandem, SignatureTranslocation)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFS = ["chr1", "chr2", "chr10"]


def load(name):
    path = os.path.join(GOLDEN, name)
    opener = gzip.open if name.endswith(".gz") else open
    with opener(path, "rt") as fh:
        return json.load(fh)


def options(d):
    o = types.SimpleNamespace(**d)
    o.genome = os.path.join(GOLDEN, d.get("genome", "ref.fa.gz"))
    return o


def row_sig(r):
    t = r[0]
    if t == "DEL":
        return SignatureDeletion(r[1], r[2], r[3], r[4], r[5])
    if t == "INS":
        return SignatureInsertion(r[1], r[2], r[3], r[4], r[5], r[6])
    if t == "INV":
        return SignatureInversion(r[1], r[2], r[3], r[4], r[5], r[6])
    if t == "DUP_TAN":
        return SignatureDuplicationTandem(r[1], r[2], r[3], r[6], r[7], r[4], r[5])
    if t == "DUP_INT":
        return SignatureInsertionFrom(r[1], r[2], r[3], r[6], r[7], r[4], r[5])
    if t == "BND":
        return SignatureTranslocation(r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8])
    raise ValueError(t)


def sig_row(s):
    t = s.type
    if t == "DEL":
        return [t, s.contig, s.start, s.end, s.signature, s.read]
    if t == "INS":
        return [t, s.contig, s.start, s.end, s.signature, s.read, s.sequence]
    if t == "INV":
        return [t, s.contig, s.start, s.end, s.signature, s.read, s.direction]
    if t == "DUP_TAN":
        return [t, s.contig, s.start, s.end, s.signature, s.read, s.copies, bool(s.fully_covered)]
    if t == "DUP_INT":
        return [t, s.contig1, s.start, s.end, s.signature, s.read, s.contig2, s.pos]
    return [t, s.contig1, s.pos1, s.direction1, s.contig2, s.pos2, s.direction2, s.signature, s.read]


def table_rows(tab, references, read_names):
    return [sig_row(s) for s in convert.objects_from_sigtable(tab, references, read_names)]


def cluster_rows(ct, references):
    """ClusterTable -> same row layout as the golden files (lists per return-tuple slot)."""
    dirs = ("fwd", "rev")
    by = [[] for _ in range(6)]
    slot = {0: 0, 1: 1, 2: 2, 3: 3, 5: 4, 4: 5}       # SVX type code -> index in the reference's return tuple
    for k in range(ct.n):
        code = int(ct.type[k])
        mem = [int(x) for x in ct.members[ct.member_off[k]:ct.member_off[k + 1]]]
        sp = None if np.isnan(ct.std_span[k]) else float(ct.std_span[k])
        po = None if np.isnan(ct.std_pos[k]) else float(ct.std_pos[k])
        if code <= 2:
            row = [references[ct.contig[k]], int(ct.start[k]), int(ct.end[k]), float(ct.score[k]), int(ct.size[k]),
                   sp, po, mem]
        else:
            row = [references[ct.contig[k]], int(ct.start[k]), int(ct.end[k]), references[ct.contig2[k]],
                   int(ct.start2[k]), int(ct.end2[k]), float(ct.score[k]), int(ct.size[k]), sp, po, mem]
            if code == 4:
                row += [dirs[ct.aux[k] & 1], dirs[(ct.aux[k] >> 1) & 1]]
        by[slot[code]].append(row)
    return by


def close(a, b, rtol=1e-9):
    if a is None or b is None:
        return a is None and b is None
    return abs(a - b) <= rtol * max(1.0, abs(b))


def compare_cluster_rows(got, exp, rtol=1e-9):
    """exact on ints / strings / member lists, rtol on the FP columns (score, std_span, std_pos)."""
    for slot in range(6):
        assert len(got[slot]) == len(exp[slot]), "slot %d: %d clusters != %d" % (slot, len(got[slot]), len(exp[slot]))
        for k, (g, e) in enumerate(zip(got[slot], exp[slot])):
            assert len(g) == len(e), (slot, k, g, e)
            for x, y in zip(g, e):
                if isinstance(y, float) or (y is None) or isinstance(x, float):
                    assert close(x, y, rtol), "slot %d cluster %d: %r != %r\n got %r\n exp %r" % (slot, k, x, y, g, e)
                else:
                    assert x == y, "slot %d cluster %d: %r != %r\n got %r\n exp %r" % (slot, k, x, y, g, e)


def sam_case_batch(case, collect_golden):
    """golden COLLECT case -> (AlignmentFile, HostBatch, options)"""
    if case.get("sam_file"):
        with open(os.path.join(GOLDEN, case["sam_file"])) as fh:
            text = fh.read()
    else:
        text = case["sam"]
        if text is None:      # all_bnds twin shares the SAM text of the preceding case with the same name/mode
            for c in collect_golden["cases"]:
                if c["name"] == case["name"] and c["mode"] == case["mode"] and c.get("sam"):
                    text = c["sam"]
                    break
    bam = records.AlignmentFile(text=text)
    o = options(case["options"])
    hb = batch.build_batch(bam, o, mode=case["mode"])
    return bam, hb, o


_C1 = {}


def c1_case():
    """BASELINE.json configs[0] input regenerated from its seed (cached per process): (golden, refs, recs)."""
    if not _C1:
        from svim_amd import synth
        g = load("g_c1.json.gz")
        refs = synth.make_reference(7, [("chr1", 2000000)])
        recs = synth.coordinate_sort(synth.planted_reads(8, 10000, refs, ["chr1"], [2000000], n_sites=300, types=("DEL", "INS"),
                                                         read_len=(1000, 6000)))
        assert len(recs) == g["n_records"] and sum(len(a.cigartuples) for a in recs) == g["n_ops"]
        _C1.update(g=g, refs=refs, recs=recs)
    return _C1["g"], _C1["refs"], _C1["recs"]


C1_FULL = dict(n_reads=10000, contig_len=250_000_000, n_sites=300, seed=1, frac_del=0.5, frac_ins=0.5, inv_read_frac=0.0,
               lengths=("triangular", 100, 20000, 15000))


def c1_full_case():
    """BASELINE.json configs[0] at the size SURVEY.md section 8(d) C1 states: 10 000 primaries of length ~ triangular(100, 20000, 15000) on one
    250 Mb contig, CIGAR = M-runs U[5,30] alternating with 1-3 bp I/D, 300 planted DEL / INS sites (log-uniform 50-5000 bp) - generated with the
    vectorised generator on the CPU (seeded torch CPU stream: the same arrays here and on the GPU box).
    -> (HostBatch, genome codes numpy uint8 [250 Mb], meta)"""
    from svim_amd import devsynth
    b, genome, meta = devsynth.make_batch(device="cpu", **C1_FULL)
    b.references = ["chr1"]
    return b.slice_records(0, b.n_rec), genome.numpy(), meta


C1_BENCH_SAMPLE = dict(n_reads=20000, n50=20000, contig_len=5_000_000, n_sites=500, seed=2)


def c1_bench_sample_case():
    """BASELINE.json configs[1] at a fiftieth of its size and the SAME densities (bench.py's workload is devsynth.make_batch(n_reads=10^6, n50=20000,
    contig_len=250 Mb, n_sites=25000): reads per Mb, planted DEL / INS / INV sites per Mb and the 12 % inversion-spanning split reads are those) - generated on
    the CPU from its seed.  -> (HostBatch with the segment table, genome codes numpy uint8, meta)"""
    from svim_amd import devsynth
    b, genome, meta = devsynth.make_batch(device="cpu", **C1_BENCH_SAMPLE)
    b.references = ["chr1"]
    return b.slice_records(0, b.n_rec), genome.numpy(), meta


def write_fasta_from_codes(path, name, codes, width=100):
    """one contig of 4-bit codes (1 2 4 8 15 = A C G T N) -> FASTA"""
    lut = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
    n = codes.shape[0]
    rows = n // width
    with open(path, "wb") as fh:
        fh.write((">%s\n" % name).encode())
        body = np.empty((rows, width + 1), dtype=np.uint8)
        body[:, :width] = lut[codes[:rows * width]].reshape(rows, width)
        body[:, width] = 10
        fh.write(body.tobytes())
        if n > rows * width:
            fh.write(lut[codes[rows * width:]].tobytes() + b"\n")


def text_close(got, exp, rtol=1e-9):
    """Two writer outputs token by token: identical except that tokens which parse as floats may differ by rtol (the FP columns:
    score, std_span, std_pos).  Returns None when equal, else a description of the first difference."""
    import re
    gl, el = got.splitlines(), exp.splitlines()
    if len(gl) != len(el):
        return "%d lines != %d" % (len(gl), len(el))
    for i, (a, b) in enumerate(zip(gl, el)):
        if a == b:
            continue
        ta, tb = re.split(r"([\t;|\[\]=])", a), re.split(r"([\t;|\[\]=])", b)
        if len(ta) != len(tb):
            return "line %d: %r != %r" % (i, a[:200], b[:200])
        for x, y in zip(ta, tb):
            if x == y:
                continue
            try:
                fx, fy = float(x), float(y)
            except ValueError:
                return "line %d: %r != %r" % (i, x, y)
            if abs(fx - fy) > rtol * max(1.0, abs(fy)):
                return "line %d: %r != %r" % (i, x, y)
    return None


def long_cigar_records():
    """(a plain record, a record whose CIGAR has > 65535 operations with one reportable insertion and one deletion, that CIGAR)"""
    import random
    from svim_amd import synth
    rng = random.Random(77)
    n_units = 33500                                   # 2 ops per unit -> 67000 ops
    cig, seq_parts = [(4, 7)], [synth.random_seq(rng, 7)]
    for u in range(n_units):
        m = rng.randint(2, 4)
        cig.append((0, m)); seq_parts.append(synth.random_seq(rng, m))
        if u == 12000:
            cig.append((1, 57)); seq_parts.append("ACGT" * 14 + "A")          # the insertion COLLECT must report, bases included
        elif u == 20000:
            cig.append((2, 120))
        else:
            op = rng.choice((1, 2))
            cig.append((op, 1))
            if op == 1:
                seq_parts.append(rng.choice("ACGT"))
    cig.append((0, 5)); seq_parts.append(synth.random_seq(rng, 5))
    assert len(cig) > 65535
    a = records.AlignedSegment()
    a.query_name, a.flag, a.reference_id, a.reference_start, a.mapping_quality = "long", 0, 0, 1000, 60
    a.cigartuples, a.query_sequence = cig, "".join(seq_parts)
    short = records.AlignedSegment()
    short.query_name, short.flag, short.reference_id, short.reference_start, short.mapping_quality = "short", 0, 0, 500, 60
    short.cigartuples, short.query_sequence = [(0, 50)], synth.random_seq(rng, 50)
    return short, a, cig


def first_json_difference(got, exp, rtol=1e-9, path="$"):
    """Two JSON-like trees: equal except that floats may differ by rtol.  None when equal, else a description of the first difference."""
    if isinstance(exp, float) or isinstance(got, float):
        if got is None or exp is None or isinstance(got, (str, bool)) or isinstance(exp, (str, bool)):
            return None if got == exp else "%s: %r != %r" % (path, got, exp)
        return None if close(got, exp, rtol) else "%s: %r != %r" % (path, got, exp)
    if isinstance(exp, dict):
        if not isinstance(got, dict) or sorted(got) != sorted(exp):
            return "%s: keys %r != %r" % (path, sorted(got) if isinstance(got, dict) else got, sorted(exp))
        for k in exp:
            d = first_json_difference(got[k], exp[k], rtol, "%s.%s" % (path, k))
            if d:
                return d
        return None
    if isinstance(exp, (list, tuple)):
        if not isinstance(got, (list, tuple)) or len(got) != len(exp):
            return "%s: %r != %r" % (path, got if not isinstance(got, (list, tuple)) else len(got), len(exp))
        for i, (a, b) in enumerate(zip(got, exp)):
            d = first_json_difference(a, b, rtol, "%s[%d]" % (path, i))
            if d:
                return d
        return None
    return None if got == exp else "%s: %r != %r" % (path, got, exp)


class ThreadAllGather(object):
    """In-process stand-in for the all-gather transport of svx_cluster_set_ranks: `world` threads meet at a barrier."""

    def __init__(self, world, timeout=120):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.slots = [None] * world
        self.calls = 0

    def for_rank(self, r):
        def allgather(send):
            self.slots[r] = bytes(send)
            self.barrier.wait()
            out = b"".join(self.slots)
            self.barrier.wait()
            self.calls += 1
            return out
        return allgather

    def abort(self):
        self.barrier.abort()


def granted_cpus():
    """CPUs this process may really use: the cgroup quota when there is one (the GPU boxes show 256 cores and grant 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def concat_batch_rows(batches):
    """record-level view of record batches (dicts of arrays in the svx_batch layout) that does not depend on how the records were cut into
    batches: (flag, tid, pos, mapq, l_seq, CIGAR bytes, packed bases, segment rows) per record"""
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
    return rows


def assert_rows_are_the_written_records(rows, recs, what=""):
    """rows of concat_batch_rows (or the same tuple layout) against the record objects a file was WRITTEN from - flag (12 bits), reference id, position, MAPQ, l_seq, the
    packed CIGAR and the bases: an assertion that two readers cannot pass by agreeing on a wrong answer"""
    assert len(rows) == len(recs), (what, len(rows), len(recs))
    lut = b"=ACMGRSVTWYHKDBN"
    for k, (x, r) in enumerate(zip(rows, recs)):
        cig = np.array([(ln << 4) | op for op, ln in (r.cigartuples or [])], dtype=np.uint32).tobytes()
        seq = r.query_sequence or ""
        assert (x[0] & 0x0fff) == r.flag and x[1] == r.reference_id and x[2] == r.reference_start and x[3] == r.mapping_quality and x[4] == len(seq) and x[5] == cig, (what, k)
        codes = np.frombuffer(x[6], dtype=np.uint8)
        nib = np.stack([codes >> 4, codes & 15], axis=1).reshape(-1)[:len(seq)]
        assert bytes(lut[int(v)] for v in nib).decode() == seq.upper(), (what, k)
