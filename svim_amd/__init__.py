"""svim_amd - MI355X-native COLLECT+CLUSTER hot path of SVIM behind SVIM's own entry points.

    from svim_amd import (analyze_alignment_file_coordsorted, analyze_alignment_file_querysorted,
                          analyze_cigar_indel, analyze_alignment_indel, analyze_read_segments,
                          cluster_sv_signatures, partition_and_cluster)

The compute runs in hand-written HIP kernels (svim_amd/csrc, gfx950) through the C ABI of include/svx.h;
there is no CPU fallback (see svim_amd/_lib.py).
"""
from .SVIM_intra import analyze_cigar_indel, analyze_alignment_indel, analyze_alignment_indel_batch   # noqa: F401
from .SVIM_inter import analyze_read_segments, analyze_read_segments_batch, is_similar      # noqa: F401
from .SVIM_COLLECT import (analyze_alignment_file_coordsorted, analyze_alignment_file_querysorted,   # noqa: F401
                           bam_iterator, retrieve_other_alignments)
from .SVIM_CLUSTER import cluster_sv_signatures                                             # noqa: F401
from .SVIM_clustering import (partition_and_cluster, form_partitions, partition_and_cluster_candidates,   # noqa: F401
                              span_position_distance_clusters, calculate_score)

__all__ = ["analyze_cigar_indel", "analyze_alignment_indel", "analyze_read_segments", "is_similar",
           "analyze_alignment_indel_batch", "analyze_read_segments_batch",
           "analyze_alignment_file_coordsorted", "analyze_alignment_file_querysorted", "bam_iterator",
           "retrieve_other_alignments", "cluster_sv_signatures", "partition_and_cluster", "form_partitions",
           "partition_and_cluster_candidates", "span_position_distance_clusters", "calculate_score"]
