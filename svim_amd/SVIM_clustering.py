"""GPU-backed counterparts of src/svim/SVIM_clustering.py (the names COMBINE / merging import stay importable)."""
import types

import numpy as np

from . import _abi, _lib, batch, convert
from ._abi import TYPE_CODE

_LABEL_TYPE = {"deleted regions": "DEL", "inserted regions": "INS", "inverted regions": "INV",
               "tandem duplicated regions": "DUP_TAN", "inserted regions with detected region of origin": "DUP_INT",
               "translocation breakpoints": "BND"}
_GENOMES = {}


def _genome_for(engine, options, contigs):
    """Upload FastaFile(options.genome) once per (engine, path, contig order)."""
    path = getattr(options, "genome", None)
    key = (id(engine), path, tuple(contigs))
    if _GENOMES.get("key") != key:
        off, codes = convert.genome_arrays(path, contigs) if path else (np.zeros(len(contigs) + 1, np.int64), np.zeros(1, np.uint8))
        engine.set_genome(off, codes)
        _GENOMES["key"] = key


def _cluster_tables(signatures, options):
    eng = _lib.engine()
    table, contigs, reads = convert.sigtable_from_objects(signatures)
    need_genome = bool((table.type == TYPE_CODE["INS"]).any())
    if need_genome:
        _genome_for(eng, options, contigs.names)
    ct = eng.cluster(_abi.Params.from_options(options), batch.contig_ranks(contigs.names), table=table)
    return ct, contigs


def cluster_signature_lists(signatures, options):
    """All six types in one device pass -> the reference's 6-tuple (DEL, INS, INV, DUP_TAN, DUP_INT, BND)."""
    ct, contigs = _cluster_tables(signatures, options)
    return convert.cluster_objects(ct, signatures, contigs.names)


def partition_and_cluster(signatures, options, type):
    """src/svim/SVIM_clustering.py:375-385 for ONE signature type (`type` is the reference's label string)."""
    if type not in _LABEL_TYPE:
        import logging
        logging.error("Unknown parameter type={0} to function partition_and_cluster.")
        return None
    want = _LABEL_TYPE[type]
    sigs = [s for s in signatures if s.type == want]
    res = cluster_signature_lists(sigs, options)
    return res[("DEL", "INS", "INV", "DUP_TAN", "DUP_INT", "BND").index(want)]


def form_partitions(sv_signatures, max_distance):
    """src/svim/SVIM_clustering.py:17-29.  Host helper kept for API compatibility (COMBINE-side callers use it on
    a handful of candidates); the GPU path partitions inside svx_cluster."""
    ordered = sorted(sv_signatures, key=lambda s: s.get_key())
    parts = []
    for s in ordered:
        if parts and parts[-1][-1].downstream_distance_to(s) <= max_distance:
            parts[-1].append(s)
        else:
            parts.append([s])
    return parts


def span_position_distance_clusters(cluster1, cluster2, position_distance_normalizer):
    """src/svim/SVIM_clustering.py:99-107 (host scalar; used by the COMBINE step on a few clusters)."""
    (_, s1, e1), (_, s2, e2) = cluster1.get_source(), cluster2.get_source()
    return abs((s1 + e1) // 2 - (s2 + e2) // 2) / position_distance_normalizer + abs((e1 - s1) - (e2 - s2)) / max(e1 - s1, e2 - s2)


def calculate_score(cluster, std_span, std_pos, span, type):
    """src/svim/SVIM_clustering.py:183-211 (host scalar helper)."""
    if std_span is None or std_pos is None:
        sds = pds = 0
    else:
        sds, pds = 1 - min(1, std_span / span), 1 - min(1, std_pos / span)
    if type == "INV":
        d = [m.direction for m in cluster]
        left = d.count("left_fwd") + d.count("left_rev")
        right = d.count("right_fwd") + d.count("right_rev")
        n = min(80, min(left, right) + d.count("all"))
    else:
        n = min(80, len(cluster))
    return n + sds * (n / 8) + pds * (n / 8)
