"""List-like views of the Structure-of-Arrays result tables that build the Python objects only when someone looks.

COLLECT and CLUSTER hand their results back as `Signature*` / `SignatureCluster*` objects (src/svim/SVSignature.py) because
the rest of SVIM - the writers (src/svim/SVIM_CLUSTER.py:29-106), COMBINE (SVIM_COMBINE.py:332-478), genotyping - consumes
objects.  Building 7*10^5 objects costs seconds of interpreter time; the GPU needs 30 ms for the same batch.  So

  * `analyze_alignment_file_*` returns SignatureList: a Sequence over the signature table.  Nothing is built until an element
    is read; the first iteration builds every object in one vectorised pass (column lists + one constructor call per row).
  * `cluster_sv_signatures` accepts a SignatureList WITHOUT touching its objects: when the list still mirrors the table
    resident on the device (same engine, same COLLECT call) the clustering runs with `source = 0/1` - nothing is uploaded -,
    otherwise the table's columns are uploaded as they are.  Any other iterable of signature objects takes the generic path
    (convert.sigtable_from_objects).
  * the six cluster lists are ClusterList views of the cluster table; a cluster's `members` resolve to signature objects on
    first access.
"""
from collections.abc import MutableSequence, Sequence

from . import convert


class SignatureList(Sequence):
    """Sequence of Signature objects backed by a SigTable (host numpy columns)."""

    def __init__(self, table, references, read_names, origin=None):
        self.table, self.references, self.read_names = table, list(references), read_names
        self.origin = origin              # (engine, collect generation, which list) while the device still holds this very table
        self._objs = None
        self._one = {}

    def __len__(self):
        return self.table.n

    def materialise(self):
        if self._objs is None:
            objs = convert.objects_from_sigtable(self.table, self.references, self.read_names)
            for i, o in self._one.items():                   # objects already handed out keep their identity
                objs[i] = o
            self._objs, self._one = objs, {}
        return self._objs

    def __iter__(self):
        return iter(self.materialise())

    def __getitem__(self, i):
        if self._objs is not None or isinstance(i, slice):
            return self.materialise()[i]
        n = self.table.n
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("signature index out of range")
        o = self._one.get(i)
        if o is None:
            if len(self._one) >= 4096:                       # many single reads: one bulk pass is cheaper from here on
                return self.materialise()[i]
            o = self._one[i] = convert.object_from_row(self.table, i, self.references, self.read_names)
        return o

    def __eq__(self, other):
        if isinstance(other, (list, tuple, SignatureList)):
            return list(self) == list(other)
        return NotImplemented

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __repr__(self):
        return "<SignatureList of %d signatures (%s)>" % (self.table.n, "objects built" if self._objs is not None else "table only")

    # what the reference's main() reports after COLLECT (src/svim/svim:117-133) without building objects
    def count_by_type(self):
        import numpy as np
        from ._abi import TYPE_NAMES
        c = np.bincount(self.table.type[:self.table.n], minlength=6)
        return {TYPE_NAMES[k]: int(c[k]) for k in range(6)}


class ClusterList(MutableSequence):
    """The clusters of ONE type (one slot of cluster_sv_signatures' 6-tuple) as a view of rows [lo, hi) of the cluster table.

    The reference hands out plain lists and its COMBINE step MUTATES them (`del insertion_signature_clusters[i]`,
    src/svim/SVIM_COMBINE.py:455-457; `translocation_signature_clusters.extend(...)`, src/svim/SVIM_merging.py:106;
    `insertion_from_signature_clusters.extend(...)`, SVIM_COMBINE.py:392), so this is a MutableSequence: the first mutation builds
    the objects and from then on the view behaves like the list it stands for (the table itself is never changed)."""

    def __init__(self, ct, lo, hi, signatures, references):
        self.ct, self.lo, self.hi, self.signatures, self.references = ct, lo, hi, signatures, references
        self._objs = None

    def __len__(self):
        return self.hi - self.lo if self._objs is None else len(self._objs)

    def materialise(self):
        if self._objs is None:
            self._objs = convert.cluster_objects_range(self.ct, self.lo, self.hi, self.signatures, self.references)
        return self._objs

    def __copy__(self):                             # copy.copy(list) is a new list of the same objects: mutating the copy leaves the original alone
        c = ClusterList(self.ct, self.lo, self.hi, self.signatures, self.references)
        c._objs = list(self.materialise())
        return c

    copy = __copy__

    def __iter__(self):
        return iter(self.materialise())

    def __getitem__(self, i):
        return self.materialise()[i]

    def __setitem__(self, i, value):
        self.materialise()[i] = value

    def __delitem__(self, i):
        del self.materialise()[i]

    def insert(self, i, value):
        self.materialise().insert(i, value)

    def extend(self, values):                       # (MutableSequence.extend appends one by one; `x.extend(x)` must not loop forever)
        self.materialise().extend(list(values))

    def sort(self, key=None, reverse=False):
        self.materialise().sort(key=key, reverse=reverse)

    def __eq__(self, other):
        if isinstance(other, (list, tuple, ClusterList)):
            return list(self) == list(other)
        return NotImplemented

    __hash__ = None

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __repr__(self):
        return "<ClusterList of %d clusters>" % len(self)
