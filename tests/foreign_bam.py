"""Test-side writer of BAM files the way OTHER programs write them (htslib / samtools, basecallers): everything the readers of svim_amd
(csrc/bamio.cpp on the host, csrc/bamdev.hip on the device) get from pysam / htslib in the reference (src/svim/SVIM_COLLECT.py:133,142-143) and never
see in the files svim_amd.records.write_bam makes.

    aux fields    every type of the SAM specification - A c C s S i I f Z H and B arrays of every sub-type -, the modified-base pair MM:Z / ML:B:C of
                  ONT / HiFi BAMs, fields before AND after SA / CG
    block layout  htslib's: the header in blocks of its own, a record that does not fit the rest of a block starts a new one (records straddle
                  blocks only when they are longer than a block: one such record spans more than two blocks); or the flat layout that cuts the
                  stream every 0xff00 bytes wherever that falls
    empty blocks  EOF markers in the middle of the stream (what concatenating BGZF pieces leaves behind)
    DEFLATE       stored blocks (level 0), level 1, fixed Huffman codes (Z_FIXED), default dynamic codes; mixed per block
    records       SEQ '*' (l_seq = 0), read names of 1 and 254 characters, base qualities

Test infrastructure only."""
import struct
import zlib

_NIB = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_B_FMT = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}


def encode_aux(items):
    """items: (tag, type, value); type B: value = (sub_type, list)"""
    out = b""
    for tag, typ, val in items:
        out += tag.encode("ascii") + typ.encode("ascii")
        if typ == "A":
            out += val.encode("ascii")
        elif typ in "cCsSiI":
            out += struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I"}[typ], val)
        elif typ == "f":
            out += struct.pack("<f", val)
        elif typ in "ZH":
            out += val.encode("ascii") + b"\0"
        elif typ == "B":
            sub, arr = val
            out += sub.encode("ascii") + struct.pack("<I", len(arr)) + struct.pack("<%d%s" % (len(arr), _B_FMT[sub]), *arr)
        else:
            raise ValueError(typ)
    return out


def record_bytes(a, aux_items, qual=None, placeholder_op=3):
    """a: svim_amd.records.AlignedSegment.  aux_items REPLACE a._tags (order kept).  A CIGAR beyond 65535 operations goes into CG:B,I behind the
    other fields unless aux_items already holds a CG entry."""
    name = a.query_name.encode("ascii") + b"\0"
    cig = list(a._cigar)
    seq = a._seq or ""
    l_seq = len(seq)
    items = list(aux_items)
    if len(cig) > 65535:
        packed = [(l << 4) | o for o, l in cig]
        if not any(t == "CG" for t, _, _ in items):
            items.append(("CG", "B", ("I", packed)))
        rlen = sum(l for o, l in cig if o in (0, 2, 3, 7, 8))
        cig = [(4, l_seq), (placeholder_op, rlen)]          # htslib writes <l_seq>S<ref_len>N; its reader (bam_tag2cigar) only looks at the first operation
    codes = [_NIB.get(c, 15) for c in seq.upper()]
    if l_seq & 1:
        codes.append(0)
    sb = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    q = bytes(qual) if qual is not None else b"\xff" * l_seq
    assert len(q) == l_seq
    core = struct.pack("<iiBBHHHiiii", a.reference_id, a.reference_start, len(name), a._mapq, 4680, len(cig), a.flag, l_seq,
                       a.next_reference_id, a.next_reference_start, a.template_length)
    body = core + name + struct.pack("<%dI" % len(cig), *[(l << 4) | o for o, l in cig]) + sb + q + encode_aux(items)
    return struct.pack("<i", len(body)) + body


def header_bytes(references, lengths, sort_order="coordinate"):
    text = ("@HD\tVN:1.6\tSO:%s\n" % sort_order + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(references, lengths)) +
            "@PG\tID:foreign\tPN:foreign\tVN:0\n").encode("ascii")
    out = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(references))
    for n, l in zip(references, lengths):
        nb = n.encode("ascii") + b"\0"
        out += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    return out


def bgzf_block(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    cd = comp.compress(payload) + comp.flush()
    bsize = len(cd) + 25
    assert bsize < 65536, bsize
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + cd +
            struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))


def write(path, references, lengths, rec_bytes, layout="htslib", deflate=((6, zlib.Z_DEFAULT_STRATEGY),), empty_every=0, block_payload=0xff00,
          sort_order="coordinate", tids=None, index=True):
    """rec_bytes: list of record byte strings (record_bytes).  deflate: (level, strategy) per block, cycled.  empty_every: an EOF marker block after
    every that many data blocks.  tids (reference id per record) + index: a .bai like records.write_bai's."""
    payloads, rec_at = [], []            # rec_at: (payload index, offset inside it) of every record start
    hdr = header_bytes(references, lengths, sort_order)
    if layout == "htslib":
        for i in range(0, len(hdr), block_payload):
            payloads.append(hdr[i:i + block_payload])
        cur = b""
        for rb in rec_bytes:
            if cur and len(cur) + len(rb) > block_payload:
                payloads.append(cur)
                cur = b""
            rec_at.append((len(payloads), len(cur)))
            p = 0
            while p < len(rb):
                take = min(block_payload - len(cur), len(rb) - p)
                cur += rb[p:p + take]
                p += take
                if len(cur) == block_payload and p < len(rb):
                    payloads.append(cur)
                    cur = b""
        if cur:
            payloads.append(cur)
    else:
        raw = hdr
        starts = []
        for rb in rec_bytes:
            starts.append(len(raw))
            raw += rb
        for i in range(0, len(raw), block_payload):
            payloads.append(raw[i:i + block_payload])
        rec_at = [(s // block_payload, s % block_payload) for s in starts]
    block_at = []
    with open(path, "wb") as fh:
        for k, pl in enumerate(payloads):
            level, strategy = deflate[k % len(deflate)]
            block_at.append(fh.tell())
            fh.write(bgzf_block(pl, level, strategy))
            if empty_every and (k + 1) % empty_every == 0:
                fh.write(EOF_BLOCK)
        end_at = fh.tell()
        fh.write(EOF_BLOCK)
    if index and tids is not None and sort_order == "coordinate":
        def voff(k):
            if k >= len(rec_at):
                return end_at << 16
            b, o = rec_at[k]
            return (block_at[b] << 16) | o
        first, last = {}, {}
        for k, tid in enumerate(tids):
            if tid >= 0:
                first.setdefault(tid, k)
                last[tid] = k + 1
        with open(path + ".bai", "wb") as fh:
            fh.write(b"BAI\1" + struct.pack("<i", len(references)))
            for t in range(len(references)):
                if t in first:
                    fh.write(struct.pack("<i", 1) + struct.pack("<Ii", 0, 1) + struct.pack("<QQ", voff(first[t]), voff(last[t])))
                else:
                    fh.write(struct.pack("<i", 0))
                fh.write(struct.pack("<i", 0))
    return len(payloads)


def decorate(rng, a, k):
    """aux fields of record k: the record's own SA (if any) in the middle of fields of every type, in an order that changes from record to record"""
    n = max(1, len(a._seq or "") // 40)
    pool = [("NM", "i", rng.randrange(0, 5000)), ("ms", "i", rng.randrange(1, 1 << 20)), ("AS", "i", -rng.randrange(1, 1 << 20)), ("tp", "A", "PS"[k & 1]),
            ("cm", "C", rng.randrange(256)), ("s1", "c", -rng.randrange(1, 128)), ("s2", "s", -rng.randrange(1, 30000)), ("rl", "S", rng.randrange(65536)),
            ("de", "f", rng.random()), ("zd", "I", rng.randrange(1 << 31, 1 << 32)), ("RG", "Z", "rg%d" % (k % 3)), ("XH", "H", "1AE301"),
            ("MM", "Z", "C+m?," + ",".join(str(rng.randrange(0, 20)) for _ in range(n)) + ";C+h?;"), ("ML", "B", ("C", [rng.randrange(256) for _ in range(n)])),
            ("mv", "B", ("c", [rng.randrange(-1, 2) for _ in range(rng.randrange(0, 60))])), ("Bs", "B", ("s", [-3, 7, 30000])), ("BS", "B", ("S", [65535, 0])),
            ("Bi", "B", ("i", [-2 ** 31, 5])), ("BI", "B", ("I", [2 ** 32 - 1])), ("Bf", "B", ("f", [0.5, -1.25, 3.0])), ("BC", "B", ("C", []))]
    rng.shuffle(pool)
    cut = rng.randrange(0, len(pool) + 1)
    items = pool[:cut]
    for tag, val in a._tags.items():
        if isinstance(val, int):
            items.append((tag, "i", val))
        elif isinstance(val, float):
            items.append((tag, "f", val))
        else:
            items.append((tag, "Z", str(val)))
    items += pool[cut:]
    return items
