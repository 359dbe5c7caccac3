"""GPU-backed counterparts of src/svim/SVIM_intra.py (same names, arguments, return shapes)."""
import types

from . import _abi, _lib, batch, convert


def analyze_cigar_indel(tuples, min_length):
    """CIGAR tuples (op, len) -> [(pos_ref, pos_read, length, "INS"|"DEL")] for indels >= min_length.
    Replaces src/svim/SVIM_intra.py:8-30; runs the same CIGAR-scan kernel COLLECT uses."""
    return _lib.engine().cigar_indel([tuple(t) for t in tuples], int(min_length))


class _OneRecordFile(object):
    """fetch()-able view of a single alignment, borrowing name lookups from the caller's bam object."""

    def __init__(self, alignment, bam):
        self._a, self._bam = alignment, bam
        self.references = list(getattr(bam, "references", []))
        if not self.references:
            # a bare duck-typed bam: only the alignment's own contig can be named
            self.references = [None] * (alignment.reference_id + 1)
            self.references[alignment.reference_id] = bam.getrname(alignment.reference_id)

    def fetch(self, until_eof=True):
        yield self._a

    def get_tid(self, name):
        return self._bam.get_tid(name)


def analyze_alignment_indel(alignment, bam, query_name, options):
    """Indel signatures of ONE alignment (src/svim/SVIM_intra.py:33-51) -> (sv_signatures, all_bnds side list)."""
    o = types.SimpleNamespace(**{k: getattr(options, k) for k in vars(options)}) if hasattr(options, "__dict__") else options
    view = _OneRecordFile(alignment, bam)
    # the record is analysed whatever its flags / MAPQ are (the reference applies those filters in its caller),
    # and its split-read analysis is not part of this entry point: present it as a supplementary record
    import copy
    rec = copy.copy(alignment)
    rec.flag = (alignment.flag & 16) | 2048
    view._a = rec
    o.min_mapq = 0
    hb = batch.build_batch(view, o, mode="coordinate")
    sig, bnd = _lib.engine().collect(hb, _abi.Params.from_options(o))
    names = [query_name]
    return (convert.objects_from_sigtable(sig, view.references, names),
            convert.objects_from_sigtable(bnd, view.references, names))


def _split_by_read(tab, objs, n_reads):
    """objects of a signature table grouped by the read they come from (table order kept inside a group)"""
    out = [[] for _ in range(n_reads)]
    for k in range(tab.n):
        out[int(tab.read_id[k])].append(objs[k])
    return out


def analyze_alignment_indel_batch(alignments, bam, query_names, options):
    """analyze_alignment_indel for MANY alignments with one launch of the COLLECT kernels (a caller that loops over reads pays the
    launch + transfer latency once, not per record): -> [(sv_signatures, all_bnds side list)] in the order of `alignments`.
    Same results as calling analyze_alignment_indel(alignments[i], bam, query_names[i], options) one by one (src/svim/SVIM_intra.py:33-51)."""
    import copy
    alignments = list(alignments)
    query_names = list(query_names)
    if len(query_names) != len(alignments):
        raise ValueError("analyze_alignment_indel_batch: one query name per alignment")
    if not alignments:
        return []
    o = types.SimpleNamespace(**{k: getattr(options, k) for k in vars(options)}) if hasattr(options, "__dict__") else options
    o.min_mapq = 0
    recs = []
    for i, a in enumerate(alignments):
        r = copy.copy(a)
        r.flag = (a.flag & 16) | 2048                   # as in analyze_alignment_indel: no flag / MAPQ filter, no split-read analysis
        r.query_name = "\x01%d" % i                      # every record is a read of its own: the read id of a signature names its record
        recs.append(r)
    view = _OneRecordFile(recs[0], bam)
    view.fetch = lambda until_eof=True: iter(recs)
    for a in alignments:                                 # (a bare duck-typed bam names only the contigs its records sit on)
        if a.reference_id >= len(view.references):
            view.references.extend([None] * (a.reference_id + 1 - len(view.references)))
        if view.references[a.reference_id] is None:
            view.references[a.reference_id] = bam.getrname(a.reference_id)
    hb = batch.build_batch(view, o, mode="coordinate")
    sig, bnd = _lib.engine().collect(hb, _abi.Params.from_options(o))
    rec_of_read = [int(nm[1:]) for nm in hb.read_names]
    names = [query_names[i] for i in rec_of_read]
    s_by = _split_by_read(sig, convert.objects_from_sigtable(sig, view.references, names), len(names))
    b_by = _split_by_read(bnd, convert.objects_from_sigtable(bnd, view.references, names), len(names))
    out = [([], []) for _ in alignments]
    for rid, i in enumerate(rec_of_read):
        out[i] = (s_by[rid], b_by[rid])
    return out
