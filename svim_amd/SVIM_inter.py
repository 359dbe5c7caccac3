"""GPU-backed counterpart of src/svim/SVIM_inter.py."""
import types

from . import _abi, _lib, batch, convert


def is_similar(chr1, start1, end1, chr2, start2, end2, span_position_treshold=0.3):
    """Host scalar helper with the reference's exact arithmetic (src/svim/SVIM_inter.py:11-21); the device
    applies the same test inside the segment kernel."""
    span1, span2 = end1 - start1, end2 - start2
    position_distance = abs((start1 + end1) // 2 - (start2 + end2) // 2) / 900
    span_distance = abs(span1 - span2) / max(span1, span2)
    return bool(chr1 == chr2 and position_distance + span_distance < span_position_treshold)


class _ReadFile(object):
    def __init__(self, records, bam):
        self._recs, self._bam = records, bam
        self.references = list(getattr(bam, "references", []))

    def fetch(self, until_eof=True):
        return iter(self._recs)

    def get_tid(self, name):
        return self._bam.get_tid(name)


def analyze_read_segments(primary, supplementaries, bam, options):
    """Split-read signatures of one read from its primary + supplementary alignments
    (src/svim/SVIM_inter.py:24-302) -> (sv_signatures, all_bnds side list)."""
    import copy
    o = types.SimpleNamespace(**vars(options))
    o.min_mapq = 0                       # the caller has filtered already (SVIM_COLLECT.py:113,154)
    # query-name mode machinery: one group = this primary + the given segments, CIGAR indels suppressed by
    # raising min_sv_size for the indel pass is NOT possible in one launch, so indels are dropped afterwards
    prim = copy.copy(primary)
    prim.flag = primary.flag & ~(4 | 256 | 2048)
    sup = []
    for s in supplementaries:
        c = copy.copy(s)
        c.query_name = prim.query_name
        c.flag = (s.flag & 16) | 2048
        sup.append(c)
    view = _ReadFile([prim] + sup, bam)
    hb = batch.build_batch(view, o, mode="queryname")
    sig, bnd = _lib.engine().collect(hb, _abi.Params.from_options(o))
    names = hb.read_names
    sigs = convert.objects_from_sigtable(sig, view.references, names)
    bnds = convert.objects_from_sigtable(bnd, view.references, names)
    keep = [i for i in range(sig.n) if sig.src[i] == 1]
    keep_b = [i for i in range(bnd.n) if bnd.src[i] == 1]
    return [sigs[i] for i in keep], [bnds[i] for i in keep_b]


def analyze_read_segments_batch(reads, bam, options):
    """analyze_read_segments for MANY reads with one launch: reads = [(primary, supplementaries)] -> [(sv_signatures, all_bnds side list)]
    in the same order, equal to the one-by-one calls (src/svim/SVIM_inter.py:24-302)."""
    import copy
    reads = list(reads)
    if not reads:
        return []
    o = types.SimpleNamespace(**vars(options))
    o.min_mapq = 0
    recs, real_names = [], []
    for i, (primary, supplementaries) in enumerate(reads):
        token = "\x01%d" % i                             # one query-name group per read, whatever the records are called
        real_names.append(primary.query_name)
        prim = copy.copy(primary)
        prim.flag = primary.flag & ~(4 | 256 | 2048)
        prim.query_name = token
        recs.append(prim)
        for s in supplementaries:
            c = copy.copy(s)
            c.query_name = token
            c.flag = (s.flag & 16) | 2048
            recs.append(c)
    view = _ReadFile(recs, bam)
    hb = batch.build_batch(view, o, mode="queryname")
    sig, bnd = _lib.engine().collect(hb, _abi.Params.from_options(o))
    read_of_id = [int(nm[1:]) for nm in hb.read_names]
    names = [real_names[i] for i in read_of_id]
    sigs = convert.objects_from_sigtable(sig, view.references, names)
    bnds = convert.objects_from_sigtable(bnd, view.references, names)
    out = [([], []) for _ in reads]
    for k in range(sig.n):
        if sig.src[k] == 1:                              # split-read signatures only (CIGAR indels belong to analyze_alignment_indel)
            out[read_of_id[int(sig.read_id[k])]][0].append(sigs[k])
    for k in range(bnd.n):
        if bnd.src[k] == 1:
            out[read_of_id[int(bnd.read_id[k])]][1].append(bnds[k])
    return out
