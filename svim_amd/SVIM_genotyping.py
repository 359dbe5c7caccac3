"""GENOTYPE on the GPU: drop-in for svim.SVIM_genotyping (src/svim/SVIM_genotyping.py) - same entry points, same results.

    genotype(candidates, bam, type, options)          mutates the candidates like the reference (:79-93)
    span_position_distance(candidate, signature, n)   (:9-31)

The reference re-fetches the BAM around every candidate (:48).  Here the alignment records are indexed once per file
(`AlignmentIndex`: reference_start / reference_end / flag / mapq / interned query name, file order) and the per-candidate walk is
an interval join on the device (`svx_genotype`, svim_amd/csrc/genotype.hip).  The candidates only have to quack like the
reference's: get_source() / get_destination(), .score, .members (signatures with .read), and receive support_fraction, genotype,
ref_reads, alt_reads.
"""
import numpy as np

from . import _abi
from ._abi import ptr


class AlignmentIndex(object):
    """Structure-of-arrays view of every alignment record of a coordinate-sorted file, in file order."""

    def __init__(self, bam):
        self.references = list(bam.references)
        self.lengths = [int(x) for x in bam.lengths]
        tid, pos, end, flag, mapq, name = [], [], [], [], [], []
        self.name_ids = {}
        for a in bam.fetch(until_eof=True):
            if a.reference_id is None or a.reference_id < 0:
                continue                                    # records without a position are never returned by a region fetch
            tid.append(a.reference_id)
            pos.append(a.reference_start)
            re = a.reference_end
            end.append(re if re is not None else a.reference_start)
            flag.append(a.flag)
            mapq.append(a.mapping_quality)
            name.append(self.name_ids.setdefault(a.query_name, len(self.name_ids)))
        tid = np.asarray(tid, dtype=np.int32)
        pos = np.asarray(pos, dtype=np.int32)
        if tid.size and ((np.diff(tid) < 0).any() or ((np.diff(tid) == 0) & (np.diff(pos) < 0)).any()):
            raise ValueError("genotyping needs a coordinate-sorted alignment file")
        n_contig = len(self.references)
        self.n, self.n_contig = int(tid.size), n_contig
        self.contig_first = np.searchsorted(tid, np.arange(n_contig + 1), side="left").astype(np.int64)
        self.contig_len = np.asarray(self.lengths, dtype=np.int64)
        self.pos = pos
        self.end = np.asarray(end, dtype=np.int32)
        self.flag = np.asarray(flag, dtype=np.uint16)
        self.mapq = np.asarray(mapq, dtype=np.uint8)
        self.name_id = np.asarray(name, dtype=np.int32)

    def view(self):
        v = _abi.AlnIndex()
        v.n, v.n_contig = self.n, self.n_contig
        pad = lambda a, dt: a if a.size else np.zeros(1, dtype=dt)      # noqa: E731
        self._keep = [self.contig_first, pad(self.contig_len, np.int64), pad(self.pos, np.int32), pad(self.end, np.int32),
                      pad(self.flag, np.uint16), pad(self.mapq, np.uint8), pad(self.name_id, np.int32)]
        v.contig_first, v.contig_len, v.pos, v.end, v.flag, v.mapq, v.name_id = [ptr(a) for a in self._keep]
        return v


_INDEX_ATTR = "_svx_alignment_index"


def alignment_index(bam):
    """The index of `bam`, built on first use and kept on the object."""
    ix = getattr(bam, _INDEX_ATTR, None)
    if ix is None:
        ix = AlignmentIndex(bam)
        try:
            setattr(bam, _INDEX_ATTR, ix)
        except AttributeError:
            pass
    return ix


def span_position_distance(candidate, signature, position_distance_normalizer):
    """src/svim/SVIM_genotyping.py:9-31."""
    c_contig, c_start, c_end = candidate.get_destination() if candidate.type in ("INS", "DUP_INT") else candidate.get_source()
    s_contig, s_start, s_end = signature.get_destination() if signature.type == "DUP_INT" else signature.get_source()
    compatible = candidate.type == signature.type or {candidate.type, signature.type} == {"INS", "DUP_INT"}
    if not compatible or c_contig != s_contig:
        return float("inf")
    span1, span2 = c_end - c_start, s_end - s_start
    center1, center2 = (c_start + c_end) // 2, (s_start + s_end) // 2
    position_distance = min(abs(c_start - s_start), abs(c_end - s_end), abs(center1 - center2)) / position_distance_normalizer
    return position_distance + abs(span1 - span2) / max(span1, span2)


def _engine(engine):
    if engine is not None:
        return engine
    from ._lib import engine as default_engine
    return default_engine()


def genotype(candidates, bam, type, options, engine=None):
    """src/svim/SVIM_genotyping.py:34-93.  `type` is "DEL", "INV", "INS" or "DUP_INT" as at src/svim/svim:164-170."""
    eng = _engine(engine)
    index = alignment_index(bam)
    if getattr(eng, "_svx_index_id", None) != id(index):
        eng.set_alignment_index(index)
        eng._svx_index_id = id(index)
    point = type in ("INS", "DUP_INT")
    todo = [c for c in candidates if not c.score < options.minimum_score]              # :38-39
    tid, start, end, moff, mnames, alt = [], [], [], [0], [], []
    name_ids = index.name_ids
    for c in todo:
        contig, s, e = c.get_destination() if point else c.get_source()               # :41-46
        if point:
            e = s
        tid.append(index.references.index(contig) if contig in index.references else -1)
        start.append(s)
        end.append(e)
        reads = set(sig.read for sig in c.members)                                     # :50
        alt.append(len(reads))
        ids = sorted(name_ids[r] for r in reads if r in name_ids)
        mnames.extend(ids)
        moff.append(len(mnames))
    ref = eng.genotype(1 if point else 0, tid, start, end, moff, np.asarray(mnames, dtype=np.int32), int(options.min_mapq)) if todo else []
    for c, n_alt, n_ref in zip(todo, alt, ref):
        n_ref = int(n_ref)
        total = n_alt + n_ref
        if total >= options.minimum_depth:                                             # :79-88
            c.support_fraction = n_alt / total
            if c.support_fraction >= options.homozygous_threshold:
                c.genotype = "1/1"
            elif c.support_fraction >= options.heterozygous_threshold:
                c.genotype = "0/1"
            elif c.support_fraction < options.heterozygous_threshold:
                c.genotype = "0/0"
            else:
                c.genotype = "./."
        elif total > 0:                                                                # :89-91
            c.support_fraction = n_alt / total
            c.genotype = "./."
        else:
            c.support_fraction = "."
            c.genotype = "./."
        c.ref_reads = n_ref
        c.alt_reads = n_alt
