"""GPU-backed counterpart of src/svim/SVIM_CLUSTER.py."""
from .SVIM_clustering import cluster_signature_lists, partition_and_cluster      # noqa: F401


def cluster_sv_signatures(sv_signatures, options):
    """Split by type, partition, cluster, consolidate (src/svim/SVIM_CLUSTER.py:7-26) - one device pass for all
    six types.  Returns (DEL, INS, INV, DUP_TAN, DUP_INT, BND) lists of SignatureCluster{UniLocal,BiLocal}."""
    return cluster_signature_lists(sv_signatures, options)


# ---- writers (src/svim/SVIM_CLUSTER.py:29-106): same files, same lines ------------------------------------------------------
# (file name, slot of the 6-tuple, which bed line(s) of a cluster go into it)
_BED_FILES = (("del.bed", 0, None), ("ins.bed", 1, None), ("inv.bed", 2, None),
              ("dup_tan_source.bed", 3, (0,)), ("dup_tan_dest.bed", 3, (1,)),
              ("trans.bed", 5, (0, 1)), ("dup_int.bed", 4, (0, 1)))
_VCF_HEADER = (
    "##fileformat=VCFv4.3", "##source=SVIMV{version}",
    '##ALT=<ID=DEL,Description="Deletion">', '##ALT=<ID=INV,Description="Inversion">', '##ALT=<ID=DUP,Description="Duplication">',
    '##ALT=<ID=DUP:TANDEM,Description="Tandem Duplication">', '##ALT=<ID=INS,Description="Insertion">',
    '##INFO=<ID=END,Number=1,Type=Integer,Description="End position of the variant described in this record">',
    '##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">',
    '##INFO=<ID=SVLEN,Number=.,Type=Integer,Description="Difference in length between REF and ALT alleles">',
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO")


def _signature_dir(working_dir):
    import os
    d = os.path.join(working_dir, "signatures")
    os.makedirs(d, exist_ok=True)
    return d


def write_signature_clusters_bed(working_dir, clusters):
    """<working_dir>/signatures/{del,ins,inv,dup_tan_source,dup_tan_dest,trans,dup_int}.bed (src/svim/SVIM_CLUSTER.py:29-70)."""
    import os
    d = _signature_dir(working_dir)
    for name, slot, which in _BED_FILES:
        with open(os.path.join(d, name), "w") as fh:
            for cluster in clusters[slot]:
                if which is None:
                    fh.write(cluster.get_bed_entry() + "\n")
                else:
                    entries = cluster.get_bed_entries()
                    for k in which:
                        fh.write(entries[k] + "\n")


def write_signature_clusters_vcf(working_dir, clusters, version):
    """<working_dir>/signatures/all.vcf: header, then the DEL / INS / INV / DUP_TAN clusters sorted by source locus
    (src/svim/SVIM_CLUSTER.py:73-106)."""
    import os
    entries = [(c.get_source(), c.get_vcf_entry()) for slot in (0, 1, 2, 3) for c in clusters[slot]]
    with open(os.path.join(_signature_dir(working_dir), "all.vcf"), "w") as fh:
        for line in _VCF_HEADER:
            fh.write(line.format(version=version) + "\n")
        for _, entry in sorted(entries, key=lambda pair: pair[0]):
            fh.write("%s\n" % entry)
