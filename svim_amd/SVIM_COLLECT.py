"""GPU-backed counterparts of src/svim/SVIM_COLLECT.py: same entry points, same return shapes.

    sv_signatures, translocation_signatures_all_bnds = analyze_alignment_file_coordsorted(bam, options)

Both results are svim_amd.lazy.SignatureList: list-like views of the signature tables that build the Signature objects on first
access (cluster_sv_signatures takes them without building any).

`bam` may be a path, our svim_amd.records.AlignmentFile, or any object with pysam's
fetch(until_eof=True) / get_tid / references (records need the pysam attribute names).
"""
import logging

from . import _abi, _lib, batch, convert, records
from .lazy import SignatureList
from .SVIM_intra import analyze_alignment_indel          # noqa: F401  (re-exported like the reference module)
from .SVIM_inter import analyze_read_segments            # noqa: F401


def bam_iterator(bam):
    """Yield (primaries, supplementaries, secondaries) per read of a query-sorted file
    (src/svim/SVIM_COLLECT.py:8-41)."""
    group = None
    name = None
    for aln in bam.fetch(until_eof=True):
        if aln.query_name != name:
            if group is not None:
                yield group
            name, group = aln.query_name, ([], [], [])
        group[2 if aln.is_secondary else (1 if aln.is_supplementary else 0)].append(aln)
    if group is not None:
        yield group


def retrieve_other_alignments(main_alignment, bam):
    """Other alignments of the read rebuilt from the SA tag (src/svim/SVIM_COLLECT.py:44-93), as
    svim_amd.records.AlignedSegment objects.  (On the GPU path the same information travels as the segment
    table of the record batch; this function exists for callers that want the objects.)"""
    if main_alignment.get_cigar_stats()[0][5] > 0:
        return []
    try:
        sa = main_alignment.get_tag("SA")
    except KeyError:
        return []
    out = []
    for tid, pos, rev, mapq, cigar in batch._parse_sa(sa, bam):
        a = records.AlignedSegment()
        a.query_name = main_alignment.query_name
        a.query_sequence = main_alignment.query_sequence
        a.flag = 2064 if rev else 2048
        a.reference_id = tid
        a.reference_start = pos
        a.mapping_quality = mapq
        a.cigartuples = cigar
        a.query_qualities = main_alignment.query_qualities
        out.append(a)
    return out


def _open(bam):
    return records.AlignmentFile(bam) if isinstance(bam, str) else bam


def _is_bam_path(bam):
    if not isinstance(bam, str):
        return False
    try:
        import gzip
        with gzip.open(bam, "rb") as fh:
            return fh.read(4) == b"BAM\1"
    except OSError:
        return False


def _run_native(path, options, mode, batch_records=200000):
    """BAM path: native reader (svim_amd/csrc/bamio.cpp) reading batch i+1 while svx_collect works on batch i; the signature lists of
    all batches accumulate in HBM (svim_amd/harness.py:BamPipeline), so the returned lists still mirror what the device holds and
    cluster_sv_signatures starts from there."""
    from .harness import BamPipeline
    eng = _lib.engine()
    pipe = BamPipeline(path, options, eng, batch_records=batch_records, mode=mode)
    try:
        n_rec = pipe.run()
        logging.info("Processed read {0}".format(n_rec))
        names = pipe.bam.read_names()
        refs = pipe.bam.references
    finally:
        pipe.close()                       # accumulation off: the accumulated lists stay resident as the last COLLECT result
    sig, bnd = eng.fetch_signatures(0), eng.fetch_signatures(1)
    return (SignatureList(sig, refs, names, origin=(eng, eng.collect_generation, 0)),
            SignatureList(bnd, refs, names, origin=(eng, eng.collect_generation, 1)))


def _run(bam, options, mode):
    if _is_bam_path(bam):
        return _run_native(bam, options, mode)
    bam = _open(bam)
    hb = batch.build_batch(bam, options, mode=mode)
    logging.info("Processed read {0}".format(hb.n_rec))
    eng = _lib.engine()
    sig, bnd = eng.collect(hb, _abi.Params.from_options(options))
    refs = hb.references
    return (SignatureList(sig, refs, hb.read_names, origin=(eng, eng.collect_generation, 0)),
            SignatureList(bnd, refs, hb.read_names, origin=(eng, eng.collect_generation, 1)))


def analyze_alignment_file_coordsorted(bam, options):
    """src/svim/SVIM_COLLECT.py:132-167 on the GPU."""
    return _run(bam, options, "coordinate")


def analyze_alignment_file_querysorted(bam, options):
    """src/svim/SVIM_COLLECT.py:96-129 on the GPU."""
    return _run(bam, options, "queryname")
