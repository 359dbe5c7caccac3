"""Seeded synthetic alignments for tests, golden-vector generation and small benchmarks.

Modelled on the reference's random-SAM test generator (src/tests/test_Collect.py:17-128: M-runs
alternating with short I/D, triangular read lengths, primaries with SA tags and hard-clipped
supplementary records) and on SURVEY.md section 8(d)'s synthetic configurations.  Host-side Python,
not on the measured path (bench.py builds its large batches directly on the GPU).
"""
import random

from .records import AlignedSegment, cigar_to_string

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(_COMP.get(c, "N") for c in reversed(s))


def random_seq(rng, n):
    return "".join(rng.choices("ACGT", k=n))


def make_reference(seed, contigs):
    """contigs: list of (name, length) -> dict name -> sequence (upper-case ACGT, a few N runs)."""
    rng = random.Random(seed)
    out = {}
    for name, length in contigs:
        s = rng.choices("ACGT", k=length)
        # sprinkle lower-case and N to exercise .upper() / non-ACGT symbols
        for _ in range(max(1, length // 20000)):
            p = rng.randrange(0, max(1, length - 50))
            for i in range(p, min(length, p + rng.randrange(5, 40))):
                s[i] = "N"
        out[name] = "".join(s)
    return out


def write_fasta(path, refs, width=60, lower_every=7):
    with open(path, "w") as fh:
        for i, (name, seq) in enumerate(refs.items()):
            fh.write(">%s some description\n" % name)
            for j in range(0, len(seq), width):
                line = seq[j:j + width]
                if lower_every and (j // width) % lower_every == 3:
                    line = line.lower()      # soft-masked stretch: fetch().upper() must undo it
                fh.write(line + "\n")


def noisy_core(rng, qlen, min_run=5, max_run=30, indel_max=3, big=None):
    """CIGAR core consuming exactly qlen query bases: M runs alternating with short I/D.
    big: optional list of (query_offset, op, length) large indels to plant (op 1 or 2)."""
    ops = []
    consumed = 0
    big = sorted(big or [])
    bi = 0
    while consumed < qlen:
        run = min(qlen - consumed, rng.randint(min_run, max_run))
        if bi < len(big) and consumed + run >= big[bi][0] > consumed:
            run = big[bi][0] - consumed
        if run > 0:
            if ops and ops[-1][0] == 0:
                ops[-1] = (0, ops[-1][1] + run)
            else:
                ops.append((rng.choice((0, 0, 0, 7, 8)) if rng.random() < 0.05 else 0, run))
            consumed += run
        if bi < len(big) and consumed == big[bi][0]:
            _, op, l = big[bi]
            bi += 1
            if op == 1:
                l = min(l, qlen - consumed)
                if l <= 0:
                    continue
                consumed += l
            ops.append((op, l))
            continue
        if consumed >= qlen:
            break
        if rng.random() < 0.5:
            l = min(rng.randint(1, indel_max), qlen - consumed - 1)
            if l > 0:
                ops.append((1, l))
                consumed += l
        else:
            ops.append((2, rng.randint(1, indel_max)))
    # a CIGAR core must not end in I/D
    while ops and ops[-1][0] in (1, 2):
        op, l = ops.pop()
        if op == 1:
            ops.append((0, l))
    merged = []
    for op, l in ops:
        if merged and merged[-1][0] == op:
            merged[-1] = (op, merged[-1][1] + l)
        else:
            merged.append((op, l))
    return merged


def ref_span(core):
    return sum(l for op, l in core if op in (0, 2, 3, 7, 8))


class Segment(object):
    def __init__(self, q_start, q_end, tid, ref_start, reverse, core, mapq):
        self.q_start, self.q_end, self.tid, self.ref_start = q_start, q_end, tid, ref_start
        self.reverse, self.core, self.mapq = reverse, core, mapq

    def clipped_cigar(self, read_len, hard=False):
        left, right = self.q_start, read_len - self.q_end
        if self.reverse:
            left, right = right, left
        c = []
        clip = 5 if hard else 4
        if left:
            c.append((clip, left))
        c.extend(self.core)
        if right:
            c.append((clip, right))
        return c


def records_for_read(name, seq, segments, references, hard_clip_suppl=True, extra_flag=0):
    """Turn a split read (forward-orientation sequence + segments) into SAM-style records:
    the longest segment becomes the primary, the rest supplementary (flag 2048), each with an SA tag."""
    L = len(seq)
    order = sorted(range(len(segments)), key=lambda i: -(segments[i].q_end - segments[i].q_start))
    recs = []
    for rank, i in enumerate(order):
        s = segments[i]
        a = AlignedSegment()
        a.query_name = name
        suppl = rank > 0
        a.flag = (16 if s.reverse else 0) | (2048 if suppl else 0) | extra_flag
        a.reference_id = s.tid
        a.reference_start = s.ref_start
        a._mapq = s.mapq
        hard = suppl and hard_clip_suppl
        a.cigartuples = s.clipped_cigar(L, hard=hard)
        full = revcomp(seq) if s.reverse else seq
        if hard:
            lo = a.cigartuples[0][1] if a.cigartuples[0][0] == 5 else 0
            hi = a.cigartuples[-1][1] if a.cigartuples[-1][0] == 5 else 0
            a.query_sequence = full[lo:L - hi]
        else:
            a.query_sequence = full
        sa = []
        for j in order:
            if j == i:
                continue
            o = segments[j]
            sa.append("%s,%d,%s,%s,%d,%d" % (references[o.tid], o.ref_start + 1, "-" if o.reverse else "+",
                                             cigar_to_string(o.clipped_cigar(L)), o.mapq, 0))
        if sa:
            a.set_tag("SA", ";".join(sa) + ";")
        recs.append(a)
    return recs


def fuzz_split_reads(seed, n_reads, references, lengths, max_sv_size=100000, read_len=(300, 2200),
                     p_noise_big=0.3):
    """Random split-read layouts chosen so that every branch of the reference's adjacent-pair
    decision tree (src/svim/SVIM_inter.py:58-240) and both post-passes (:242-300) are reachable."""
    rng = random.Random(seed)
    recs = []
    for r in range(n_reads):
        L = rng.randint(*read_len)
        seq = random_seq(rng, L)
        nseg = rng.choice((1, 1, 2, 2, 2, 3, 3, 4, 5, 6))
        # cut points on the read with gaps/overlaps
        cuts = sorted(rng.sample(range(40, L - 40), nseg - 1)) if nseg > 1 else []
        bounds = [0] + cuts + [L]
        segs = []
        tid = rng.randrange(len(references))
        rev = rng.random() < 0.4
        ref_cursor = rng.randint(1000, max(2000, lengths[tid] // 2))
        lead = rng.choice((0, 0, rng.randint(1, 60)))
        for k in range(nseg):
            qs, qe = bounds[k], bounds[k + 1]
            if k == 0:
                qs += lead
            else:
                qs += rng.choice((0, 0, 0, rng.randint(-8, 12), rng.randint(-8, 12), rng.randint(13, 400)))
            qs = max(0, min(qs, qe - 20))
            if k == nseg - 1:
                qe -= rng.choice((0, 0, rng.randint(1, 60)))
            big = None
            if rng.random() < p_noise_big and qe - qs > 200:
                big = [(rng.randint(30, qe - qs - 100), rng.choice((1, 2)), rng.choice((39, 40, 41, 80, 250)))]
            core = noisy_core(rng, qe - qs, big=big)
            span = ref_span(core)
            if k > 0:
                prev = segs[-1]
                scen = rng.random()
                new_tid, new_rev = prev.tid, prev.reverse
                if scen < 0.12:
                    new_tid = rng.choice([t for t in range(len(references)) if t != prev.tid] or [prev.tid])
                if 0.08 < scen < 0.38:
                    new_rev = not prev.reverse
                jump = rng.choice((
                    rng.randint(-6, 12),                 # adjacent (INS if read gap) / nothing
                    rng.randint(35, 45),                 # around min_sv_size
                    rng.randint(40, 3000),               # DEL-sized
                    rng.randint(max_sv_size - 50, max_sv_size + 50),
                    rng.randint(max_sv_size + 1, max_sv_size + 40000),   # BND
                    -rng.randint(35, 45),
                    -rng.randint(40, 3000),              # tandem dup
                    -rng.randint(3000, 12000),           # large tandem
                    -rng.randint(max_sv_size + 1, max_sv_size + 30000),
                ))
                if new_tid != prev.tid:
                    start = rng.randint(1000, lengths[new_tid] - span - 1000)
                elif new_rev == prev.reverse:
                    if not new_rev:
                        start = prev.ref_start + ref_span(prev.core) + jump
                    else:
                        start = prev.ref_start - jump - span
                else:
                    # inversion-like geometry: four cases keyed on relative position
                    pe = prev.ref_start + ref_span(prev.core)
                    which = rng.randrange(4)
                    if which == 0:
                        start = pe + rng.randint(-5, 3000)
                    elif which == 1:
                        start = prev.ref_start - span - rng.randint(-5, 3000)
                    elif which == 2:
                        start = pe + rng.randint(max_sv_size, max_sv_size + 20000)
                    else:
                        start = prev.ref_start - span - rng.randint(max_sv_size, max_sv_size + 20000)
                start = max(10, min(start, lengths[new_tid] - span - 10))
                tid, rev = new_tid, new_rev
            else:
                start = max(10, min(ref_cursor, lengths[tid] - span - 10))
            mapq = rng.choice((60, 60, 60, 60, 30, 20, 19, 5))
            segs.append(Segment(qs, qe, tid, start, rev, core, mapq))
        # occasionally build a cut&paste / interspersed duplication layout: A -> (far B) -> A'
        if nseg >= 3 and rng.random() < 0.35:
            a, b, c = segs[0], segs[1], segs[2]
            b.q_start, c.q_start = a.q_end + rng.randint(-3, 5), None
            b.q_end = max(b.q_start + 30, b.q_end)
            b.core = noisy_core(rng, b.q_end - b.q_start)
            c.q_start = b.q_end + rng.randint(-3, 5)
            c.q_end = max(c.q_start + 30, c.q_end)
            c.core = noisy_core(rng, c.q_end - c.q_start)
            for s in (b, c):
                s.reverse = a.reverse
            b.tid = rng.choice(range(len(references)))
            far = rng.randint(max_sv_size + 5000, max_sv_size + 60000)
            if b.tid == a.tid:
                b.ref_start = max(10, min(a.ref_start + far * rng.choice((-1, 1)), lengths[b.tid] - ref_span(b.core) - 10))
            else:
                b.ref_start = rng.randint(1000, lengths[b.tid] - ref_span(b.core) - 1000)
            c.tid = a.tid
            if not a.reverse:
                c.ref_start = a.ref_start + ref_span(a.core) + rng.randint(-20, 20)
            else:
                c.ref_start = a.ref_start - ref_span(c.core) + rng.randint(-20, 20)
            c.ref_start = max(10, min(c.ref_start, lengths[c.tid] - ref_span(c.core) - 10))
            # keep the query ends consistent
            for s in segs[3:]:
                s.q_start = max(s.q_start, c.q_end)
                if s.q_end - s.q_start < 20:
                    s.q_end = s.q_start + 20
                s.core = noisy_core(rng, s.q_end - s.q_start)
            need = max(s.q_end for s in segs)
            if need > L:
                seq = seq + random_seq(rng, need - L)
                L = need
        segs = [s for s in segs if s.q_end <= L and s.q_end - s.q_start >= 1]
        extra = 0
        roll = rng.random()
        if roll < 0.03:
            extra = 256           # secondary
        elif roll < 0.05:
            extra = 4             # unmapped
        recs.extend(records_for_read("read%d" % r, seq, segs, references,
                                     hard_clip_suppl=rng.random() < 0.7, extra_flag=extra))
    return recs


def inversion_insertion_layouts(seed, n_reads, references, lengths, max_sv_size=100000):
    """Two-segment split reads aimed at the branches the random fuzz reaches rarely: the four inversion geometries of
    src/svim/SVIM_inter.py:152-204 (left_fwd / left_rev / right_fwd / right_rev) and split-read insertions on both strands
    (:78-92), with sizes around min_sv_size, in the bulk, and at max_sv_size."""
    rng = random.Random(seed)
    recs = []
    kinds = ("inv_left_fwd", "inv_left_rev", "inv_right_fwd", "inv_right_rev", "ins_fwd", "ins_rev")
    for r in range(n_reads):
        kind = kinds[r % len(kinds)]
        size = rng.choice((38, 39, 40, 41, 60, 150, 700, 2500, max_sv_size - 1, max_sv_size, max_sv_size + 1))
        tid = rng.randrange(len(references))
        l1, l2 = rng.randint(120, 900), rng.randint(120, 900)
        if kind in ("inv_left_rev", "inv_right_fwd"):
            l1 = max(45, min(l1, size))
        if kind in ("inv_left_fwd", "inv_right_rev"):
            l2 = max(45, min(l2, size))
        gap = size if kind.startswith("ins") else rng.choice((0, 0, 0, -3, 6, 12))
        lead, trail = rng.choice((0, rng.randint(1, 40))), rng.choice((0, rng.randint(1, 40)))
        c1, c2 = noisy_core(rng, l1), noisy_core(rng, l2)
        s1, s2 = ref_span(c1), ref_span(c2)
        p = rng.randint(max_sv_size + 5000, max(max_sv_size + 6000, lengths[tid] - max_sv_size - 5000)) if lengths[tid] > 2 * max_sv_size + 12000 \
            else rng.randint(3000, max(3001, lengths[tid] - 3000 - s1 - s2))
        jit = rng.randint(-4, 4)
        if kind == "inv_left_fwd":          # fwd [p, p+s1) then rev ending at cur.ref_end + size
            a_rev, b_rev, a_start, b_start = False, True, p, p + s1 + size - s2
        elif kind == "inv_left_rev":        # fwd then rev lying to its LEFT: INV(next.ref_end, cur.ref_end)
            a_rev, b_rev, a_start, b_start = False, True, p, p + s1 - size - s2
        elif kind == "inv_right_fwd":       # rev then fwd to its right: INV(cur.ref_start, next.ref_start)
            a_rev, b_rev, a_start, b_start = True, False, p, p + size
        elif kind == "inv_right_rev":       # rev then fwd to its left: INV(next.ref_start, cur.ref_start)
            a_rev, b_rev, a_start, b_start = True, False, p, p - size
        elif kind == "ins_fwd":             # same strand, adjacent on the reference, `size` unaligned read bases in between
            a_rev, b_rev, a_start, b_start = False, False, p, p + s1 + jit
        else:                               # reverse strand: the later read part maps to the LEFT
            a_rev, b_rev, a_start, b_start = True, True, p, p - jit - s2
        b_start = max(5, min(b_start, lengths[tid] - s2 - 5))
        a_start = max(5, min(a_start, lengths[tid] - s1 - 5))
        q1s, q1e = lead, lead + l1
        q2s = max(q1e + gap, q1s + 1)
        q2e = q2s + l2
        L = q2e + trail
        seq = random_seq(rng, L)
        segs = [Segment(q1s, q1e, tid, a_start, a_rev, c1, 60), Segment(q2s, q2e, tid, b_start, b_rev, c2, rng.choice((60, 60, 60, 25)))]
        recs.extend(records_for_read("lay%d" % r, seq, segs, references, hard_clip_suppl=rng.random() < 0.5))
    return recs


def planted_reads(seed, n_reads, refs, references, lengths, n_sites=40, types=("DEL", "INS"),
                  read_len=(2000, 12000), tid=0, size_range=(50, 2000), err=0.03):
    """Reads over contig `tid` carrying planted DEL/INS (in CIGAR) and INV (as fwd-rev-fwd split
    reads with SA tags) sites, with +-jitter in size/position."""
    rng = random.Random(seed)
    clen = lengths[tid]
    sites = []
    for _ in range(n_sites):
        typ = rng.choice(types)
        size = int(round(size_range[0] * (size_range[1] / size_range[0]) ** rng.random()))
        pos = rng.randint(3000, clen - 3000 - size)
        hom = rng.random() < 0.5
        ins_seq = random_seq(rng, size) if typ == "INS" else None
        sites.append((pos, typ, size, hom, ins_seq))
    sites.sort()
    recs = []
    ref = refs[references[tid]]
    for r in range(n_reads):
        L = int(rng.triangular(read_len[0], read_len[1], read_len[1] * 0.75))
        start = rng.randint(100, clen - L - 3000)
        name = "pread%d" % r
        hap = rng.random() < 0.5
        inside = [s for s in sites if start + 200 < s[0] < start + L - 200 - (s[2] if s[1] != "INS" else 0)
                  and (s[3] or hap)]
        inv = [s for s in inside if s[1] == "INV"]
        if inv:
            s = rng.choice(inv)
            a, b = s[0] + rng.randint(-5, 5), s[0] + s[2] + rng.randint(-5, 5)
            q1 = a - start
            q2 = q1 + (b - a)
            if q2 + 100 < L and q1 > 100:
                seq = random_seq(rng, L)
                segs = [Segment(0, q1, tid, start, False, noisy_core(rng, q1), 60)]
                c2 = noisy_core(rng, q2 - q1)
                segs.append(Segment(q1, q2, tid, b - ref_span(c2), True, c2, 60))
                c3 = noisy_core(rng, L - q2)
                segs.append(Segment(q2, L, tid, b, False, c3, 60))
                segs[0].ref_start = a - ref_span(segs[0].core)
                recs.extend(records_for_read(name, seq, segs, references, hard_clip_suppl=True))
                continue
        big = []
        seq_parts = {}
        for pos, typ, size, hom, ins_seq in inside:
            if typ not in ("DEL", "INS"):
                continue
            off = pos - start + rng.randint(-10, 10)
            sz = max(1, size + rng.randint(-5, 5))
            if off <= 30 or off >= L - sz - 30 or (big and off <= big[-1][0] + big[-1][2] + 60):
                continue
            big.append((off, 2 if typ == "DEL" else 1, sz))
            if typ == "INS":
                noisy = [c if rng.random() > err else rng.choice("ACGT") for c in (ins_seq * 2)[:sz]]
                seq_parts[off] = "".join(noisy)
        core = noisy_core(rng, L, big=big)
        # build the read sequence along the CIGAR
        out = []
        rp = start
        qp = 0
        for op, l in core:
            if op in (0, 7, 8):
                out.append(ref[rp:rp + l].replace("N", "A") if rp + l <= clen else random_seq(rng, l))
                rp += l
                qp += l
            elif op == 1:
                out.append(seq_parts.get(qp, None) if (qp in seq_parts and len(seq_parts[qp]) == l) else random_seq(rng, l))
                qp += l
            elif op == 2:
                rp += l
        seq = "".join(out)
        seq = (seq + random_seq(rng, L))[:L]
        a = AlignedSegment()
        a.query_name = name
        a.flag = 0
        a.reference_id = tid
        a.reference_start = start
        a._mapq = rng.choice((60, 60, 60, 40, 10))
        lead = rng.choice((0, 0, rng.randint(1, 50)))
        cig = list(core)
        if lead:
            cig = [(4, lead)] + cig
            seq = random_seq(rng, lead) + seq
        a.cigartuples = cig
        a.query_sequence = seq
        recs.append(a)
    return recs


def coordinate_sort(recs):
    """Stable sort like `samtools sort` (tid, pos); unmapped (tid -1) last."""
    return sorted(recs, key=lambda a: ((a.reference_id if a.reference_id >= 0 else 1 << 30), a.reference_start))


def sam_text(references, lengths, recs, sort_order="coordinate"):
    lines = ["@HD\tVN:1.6\tSO:%s" % sort_order]
    for n, l in zip(references, lengths):
        lines.append("@SQ\tSN:%s\tLN:%d" % (n, l))
    for a in recs:
        tags = []
        for k, v in a.get_tags():
            if isinstance(v, int):
                tags.append("%s:i:%d" % (k, v))
            else:
                tags.append("%s:Z:%s" % (k, v))
        f = [a.query_name, str(a.flag), references[a.reference_id] if a.reference_id >= 0 else "*",
             str(a.reference_start + 1), str(a.mapping_quality), a.cigarstring or "*", "*", "0", "0",
             a.query_sequence or "*", "*"] + tags
        lines.append("\t".join(f))
    return "\n".join(lines) + "\n"


def genotype_rows(seed, lengths, n_reads=2600, hot=((0, 20000, 700),)):
    """Alignment rows [name, flag, tid, pos, mapq, ref_len] for the GENOTYPE tests: plain one-op alignments, a share of secondary /
    unmapped-but-placed / low-mapq / supplementary records, `hot` = (tid, centre, extra reads) piles deep enough for the
    500-alignment cap of the reference's loop.  Coordinate-sorted (stable)."""
    rng = random.Random(seed)
    rows = []
    k = 0

    def add(tid, pos, ln):
        nonlocal k
        name = "g%05d" % k
        k += 1
        u = rng.random()
        flag = 16 if rng.random() < 0.5 else 0
        mapq = 60
        if u < 0.05:
            flag |= 256
        elif u < 0.08:
            flag |= 4
        elif u < 0.16:
            mapq = rng.choice((0, 5, 19))
        elif u < 0.20:
            mapq = 20
        rows.append([name, flag, tid, pos, mapq, ln])
        if rng.random() < 0.12:                       # a second record of the same read (supplementary) close by
            p2 = max(0, min(lengths[tid] - 50, pos + rng.randint(-3000, 3000)))
            rows.append([name, (flag & 16) | 2048, tid, p2, 60, rng.randint(200, 4000)])

    for _ in range(n_reads):
        tid = rng.randrange(len(lengths))
        ln = rng.randint(300, 9000)
        add(tid, rng.randint(0, max(1, lengths[tid] - ln - 1)), min(ln, lengths[tid] - 1))
    for tid, centre, extra in hot:
        for _ in range(extra):
            ln = rng.randint(1500, 9000)
            add(tid, max(0, centre - rng.randint(0, ln)), ln)
    rows.sort(key=lambda r: (r[2], r[3]))
    return rows


def genotype_sam_text(references, lengths, rows):
    head = ["@HD\tVN:1.6\tSO:coordinate"] + ["@SQ\tSN:%s\tLN:%d" % (n, l) for n, l in zip(references, lengths)]
    body = ["%s\t%d\t%s\t%d\t%d\t%dM\t*\t0\t0\t*\t*" % (r[0], r[1], references[r[2]], r[3] + 1, r[4], r[5]) for r in rows]
    return "\n".join(head + body) + "\n"
