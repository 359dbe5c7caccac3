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
