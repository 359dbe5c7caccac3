"""Host-side signature / cluster objects handed back to SVIM's downstream code.

The GPU path works on Structure-of-Arrays tables (include/svx.h); COMBINE, genotyping and the writers of
the reference (src/svim/SVIM_CLUSTER.py:29-106, SVIM_COMBINE.py:332, SVCandidate.py:98-245) expect Python
objects with the attribute and method names of src/svim/SVSignature.py:3-311.  These classes provide that
surface - same constructor argument order, attributes, get_source/get_destination/get_key/
downstream_distance_to/as_string and the BED/VCF line formats - implemented table-driven rather than by
one hand-written method per class.
"""
import logging

_INF = float("inf")


class Signature(object):
    """Common behaviour.  Subclasses set `type` and may override the two locus accessors."""
    type = None
    _extra_tag = ()          # attribute names spliced into the 4th as_string column after the type
    _bilocal = False

    def __init__(self, contig, start, end, signature, read):
        self.contig, self.start, self.end, self.signature, self.read = contig, start, end, signature, read
        self.type = None
        if end < start:
            logging.warning("Signature with invalid coordinates (end < start): " + self.as_string())

    # -- loci ---------------------------------------------------------------------------------
    def get_source(self):
        return (self.contig, self.start, self.end)

    def _key_locus(self):
        """(contigs..., coordinate) the partition sort uses after the type."""
        c, s, e = self.get_source()
        return (c, e)

    def _gap_from(self):
        """coordinate the partition gap is measured from (this element as predecessor)."""
        return self.get_source()[2]

    def _gap_to(self):
        return self.get_source()[1]

    def _gap_group(self):
        return (self.type, self.get_source()[0])

    def get_key(self):
        return (self.type,) + self._key_locus()

    def downstream_distance_to(self, signature2):
        if self._gap_group() == signature2._gap_group():
            return max(0, signature2._gap_to() - self._gap_from())
        return _INF

    # -- text ---------------------------------------------------------------------------------
    def _tag(self):
        parts = [str(self.type)]
        for name, where in self._extra_tag:
            if where == "pre":
                parts.append(str(getattr(self, name)))
        parts.append(self.signature)
        for name, where in self._extra_tag:
            if where == "post":
                parts.append(str(getattr(self, name)))
        return ";".join(parts)

    def as_string(self, sep="\t"):
        if self._bilocal:
            sc, ss, se = self.get_source()
            dc, ds, de = self.get_destination()
            cols = ["%s:%s-%s" % (sc, ss, se), "%s:%s-%s" % (dc, ds, de), self._tag(), str(self.read)]
        else:
            c, s, e = self.get_source()
            cols = [str(c), str(s), str(e), self._tag(), str(self.read)]
        return sep.join(cols)

    def __repr__(self):
        return "<%s %s>" % (type(self).__name__, self.as_string(" "))


class SignatureDeletion(Signature):
    def __init__(self, contig, start, end, signature, read):
        assert end >= start
        self.contig, self.start, self.end, self.signature, self.read = contig, start, end, signature, read
        self.type = "DEL"


class SignatureInsertion(Signature):
    def __init__(self, contig, start, end, signature, read, sequence):
        assert end >= start
        self.contig, self.start, self.end, self.signature, self.read = contig, start, end, signature, read
        self.sequence = sequence
        self.type = "INS"

    def _key_locus(self):
        c, s, e = self.get_source()
        return (c, s)

    def _gap_from(self):
        return self.get_source()[1]


class SignatureInversion(Signature):
    _extra_tag = (("direction", "pre"),)

    def __init__(self, contig, start, end, signature, read, direction):
        assert end >= start
        self.contig, self.start, self.end, self.signature, self.read = contig, start, end, signature, read
        self.direction = direction
        self.type = "INV"


class SignatureInsertionFrom(Signature):
    _bilocal = True

    def __init__(self, contig1, start, end, contig2, pos, signature, read):
        assert end >= start
        self.contig1, self.start, self.end, self.contig2, self.pos = contig1, start, end, contig2, pos
        self.signature, self.read = signature, read
        self.type = "DUP_INT"

    def get_source(self):
        return (self.contig1, self.start, self.end)

    def get_destination(self):
        return (self.contig2, self.pos, self.pos + (self.end - self.start))

    def _key_locus(self):
        return (self.contig2, self.contig1, self.pos)

    def _gap_from(self):
        return self.pos

    def _gap_to(self):
        return self.pos

    def _gap_group(self):
        return (self.type, self.contig2, self.contig1)


class SignatureDuplicationTandem(Signature):
    _bilocal = True
    _extra_tag = (("copies", "post"),)

    def __init__(self, contig, start, end, copies, fully_covered, signature, read):
        assert end >= start
        self.contig, self.start, self.end, self.signature, self.read = contig, start, end, signature, read
        self.copies, self.fully_covered = copies, fully_covered
        self.type = "DUP_TAN"

    def get_destination(self):
        c, s, e = self.get_source()
        return (c, e, e + self.copies * (e - s))


class SignatureTranslocation(Signature):
    _bilocal = True

    def __init__(self, contig1, pos1, direction1, contig2, pos2, direction2, signature, read):
        # canonical orientation: the smaller (contig name, position) end comes first; swapping the
        # ends mirrors both directions
        if (contig1, pos1) < (contig2, pos2):
            self.contig1, self.pos1, self.direction1 = contig1, pos1, direction1
            self.contig2, self.pos2, self.direction2 = contig2, pos2, direction2
        else:
            flip = {"rev": "fwd"}
            self.contig1, self.pos1, self.direction1 = contig2, pos2, flip.get(direction2, "rev")
            self.contig2, self.pos2, self.direction2 = contig1, pos1, flip.get(direction1, "rev")
        self.signature, self.read = signature, read
        self.type = "BND"

    def get_source(self):
        return (self.contig1, self.pos1, self.pos1 + 1)

    def get_destination(self):
        return (self.contig2, self.pos2, self.pos2 + 1)

    def _key_locus(self):
        return (self.contig1, self.pos1)


def _members_text(members):
    return "[" + "][".join(m.as_string("|") for m in members) + "]"


class _LazyMembers(object):
    """`members` of a cluster: a list of signature objects, or - until first read - (signature sequence, index array) as the GPU
    path hands it over (svim_amd/convert.py:cluster_objects_range)."""

    @property
    def members(self):
        m = self._members
        if type(m) is tuple and len(m) == 2 and hasattr(m[1], "dtype"):
            sigs, idx = m
            m = self._members = [sigs[int(j)] for j in idx]
        return m

    @members.setter
    def members(self, value):
        self._members = value


class SignatureClusterUniLocal(_LazyMembers, Signature):
    def __init__(self, contig, start, end, score, size, members, type, std_span, std_pos):
        self.contig, self.start, self.end = contig, start, end
        self.score, self.size, self.members, self.type = score, size, members, type
        self.std_span, self.std_pos = std_span, std_pos

    def get_bed_entry(self):
        name = "%s;%s;%s;%s" % (self.type, self.size, self.std_span, self.std_pos)
        return "\t".join(str(x) for x in (self.contig, self.start, self.end, name, self.score,
                                          _members_text(self.members)))

    def get_vcf_entry(self):
        if self.type not in ("DEL", "INS", "INV"):
            return None
        info = "SVTYPE=%s;END=%s;SVLEN=%s;STD_SPAN=%s;STD_POS=%s" % (self.type, self.end, self.end - self.start,
                                                                     self.std_span, self.std_pos)
        return "\t".join(str(x) for x in (self.contig, self.start + 1, ".", "N", "<" + self.type + ">", ".", "PASS",
                                          info))

    def get_length(self):
        return self.end - self.start


class SignatureClusterBiLocal(_LazyMembers, Signature):
    def __init__(self, source_contig, source_start, source_end, dest_contig, dest_start, dest_end, score, size,
                 members, type, std_span, std_pos):
        self.source_contig, self.source_start, self.source_end = source_contig, source_start, source_end
        self.dest_contig, self.dest_start, self.dest_end = dest_contig, dest_start, dest_end
        self.score, self.size, self.members, self.type = score, size, members, type
        self.std_span, self.std_pos = std_span, std_pos

    def get_source(self):
        return (self.source_contig, self.source_start, self.source_end)

    def get_destination(self):
        return (self.dest_contig, self.dest_start, self.dest_end)

    def get_bed_entries(self):
        mem = _members_text(self.members)
        src_name = "%s_source;%s:%s-%s;%s;%s;%s" % (self.type, self.dest_contig, self.dest_start, self.dest_end,
                                                    self.size, self.std_span, self.std_pos)
        dst_name = "%s_dest;%s:%s-%s;%s" % (self.type, self.source_contig, self.source_start, self.source_end,
                                            self.size)
        src = "\t".join(str(x) for x in (self.source_contig, self.source_start, self.source_end, src_name,
                                         self.score, mem))
        dst = "\t".join(str(x) for x in (self.dest_contig, self.dest_start, self.dest_end, dst_name, self.score, mem))
        return (src, dst)

    def get_vcf_entry(self):
        if self.type != "DUP_TAN":
            return None
        info = "SVTYPE=%s;END=%s;SVLEN=%s;STD_SPAN=%s;STD_POS=%s" % ("DUP:TANDEM", self.source_end,
                                                                     self.source_end - self.source_start,
                                                                     self.std_span, self.std_pos)
        return "\t".join(str(x) for x in (self.source_contig, self.source_start + 1, ".", "N", "<DUP:TANDEM>", ".",
                                          "PASS", info))

    def get_source_length(self):
        return self.source_end - self.source_start

    def get_destination_length(self):
        return self.dest_end - self.dest_start
