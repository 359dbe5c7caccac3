"""GPU-backed counterpart of src/svim/SVIM_CLUSTER.py."""
from .SVIM_clustering import cluster_signature_lists, partition_and_cluster      # noqa: F401


def cluster_sv_signatures(sv_signatures, options):
    """Split by type, partition, cluster, consolidate (src/svim/SVIM_CLUSTER.py:7-26) - one device pass for all
    six types.  Returns (DEL, INS, INV, DUP_TAN, DUP_INT, BND) lists of SignatureCluster{UniLocal,BiLocal}."""
    return cluster_signature_lists(list(sv_signatures), options)
