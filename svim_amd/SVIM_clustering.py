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
    """-> (ClusterTable, contig names, the sequence cluster members index into)"""
    from .lazy import SignatureList
    eng = _lib.engine()
    p = _abi.Params.from_options(options)
    if isinstance(signatures, SignatureList):
        # a COLLECT result: its table is used as it is - no objects are built, and when the context still holds this very table
        # (same engine, no svx_collect since) nothing is uploaded either
        table, names = signatures.table, signatures.references
        if bool((table.type[:table.n] == TYPE_CODE["INS"]).any()):
            _genome_for(eng, options, names)
        o = signatures.origin
        if o is not None and o[0] is eng and o[1] == eng.collect_generation:
            ct = eng.cluster(p, batch.contig_ranks(names), source=o[2])
        else:
            ct = eng.cluster(p, batch.contig_ranks(names), table=table)
        return ct, names, signatures
    signatures = list(signatures)
    table, contigs, reads = convert.sigtable_from_objects(signatures)
    if bool((table.type == TYPE_CODE["INS"]).any()):
        _genome_for(eng, options, contigs.names)
    ct = eng.cluster(p, batch.contig_ranks(contigs.names), table=table)
    return ct, contigs.names, signatures


def cluster_signature_lists(signatures, options):
    """All six types in one device pass -> the reference's 6-tuple (DEL, INS, INV, DUP_TAN, DUP_INT, BND) of lazy cluster lists."""
    ct, names, members_from = _cluster_tables(signatures, options)
    return convert.cluster_objects(ct, members_from, names)


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


def span_position_distance_intdup_candidates(signature1, signature2, position_distance_normalizer):
    """src/svim/SVIM_clustering.py:110-119 (host scalar, same operation order)."""
    (_, s1, e1), (_, s2, e2) = signature1.get_source(), signature2.get_source()
    span1, span2 = e1 - s1, e2 - s2
    pd_source = abs((s1 + e1) // 2 - (s2 + e2) // 2) / position_distance_normalizer
    pd_dest = abs(signature1.get_destination()[1] - signature2.get_destination()[1]) / position_distance_normalizer
    return pd_source + pd_dest + abs(span1 - span2) / max(span1, span2)


def partition_and_cluster_candidates(candidates, options, type):
    """COMBINE-step re-clustering of interspersed-duplication candidates (src/svim/SVIM_clustering.py:306-372, called at
    src/svim/SVIM_COMBINE.py:476).  The candidate set is small (O(#clusters)): partitioning, the seeded down-sampling and
    the pair distances are host scalars with the reference's arithmetic; average linkage + flat cut of all partitions run
    as one batch of the LDS linkage kernel (svx_linkage_fcluster).  The merged candidates are built with the class of the
    inputs (svim's CandidateDuplicationInterspersed or any class with that constructor)."""
    import logging
    from random import seed, sample
    from statistics import mean
    partitions = form_partitions(candidates, options.partition_max_distance)
    clusters = [None] * len(partitions)
    problems, where = [], []
    seed(1524)
    for k, partition in enumerate(partitions):
        if len(partition) == 1:
            clusters[k] = [[partition[0]]]
            continue
        part = sample(partition, 100) if len(partition) > 100 else partition
        d = [span_position_distance_intdup_candidates(part[i], part[j], options.position_distance_normalizer)
             for i in range(len(part) - 1) for j in range(i + 1, len(part))]
        problems.append((len(part), d))
        where.append((k, part))
    if problems:
        for (k, part), labels in zip(where, _lib.engine().linkage_fcluster(problems, float(options.cluster_max_distance))):
            groups = [[] for _ in range(int(labels.max()))]
            for idx, lab in enumerate(labels):
                groups[int(lab) - 1].append(part[idx])
            clusters[k] = groups
    flat = [c for groups in clusters for c in groups]
    logging.info("Clustered {0}: {1} partitions and {2} clusters".format(type, len(partitions), len(flat)))
    final = []
    for cluster in flat:
        if cluster[0].type != "DUP_INT":
            continue
        stds_span = [c.std_span for c in cluster if c.std_span is not None]
        stds_pos = [c.std_pos for c in cluster if c.std_pos is not None]
        n = len(cluster)
        cls = cluster[0].__class__
        final.append(cls(cluster[0].get_source()[0], int(round(sum(c.get_source()[1] for c in cluster) / n)),
                         int(round(sum(c.get_source()[2] for c in cluster) / n)), cluster[0].get_destination()[0],
                         int(round(sum(c.get_destination()[1] for c in cluster) / n)),
                         int(round(sum(c.get_destination()[2] for c in cluster) / n)),
                         [m for c in cluster for m in c.members], max(c.score for c in cluster),
                         mean(stds_span) if stds_span else None, mean(stds_pos) if stds_pos else None,
                         any(c.cutpaste for c in cluster)))
    return final
