"""Minimal BAM reader for the tests (BGZF = concatenated gzip members; SAM/BAM v1 layout)."""
import gzip
import struct


def read_bam(path):
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"BAM\x01"
    (l_text,) = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, p)
        p += 4
        name = data[p:p + l_name - 1].decode()
        p += l_name
        (l_ref,) = struct.unpack_from("<i", data, p)
        p += 4
        refs.append((name, l_ref))
    recs = []
    code = "=ACMGRSVTWYHKDBN"
    while p < len(data):
        (bs,) = struct.unpack_from("<i", data, p)
        ref_id, pos, l_rn, mapq, bin_, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", data, p + 4)
        q = p + 36
        name = data[q:q + l_rn - 1].decode()
        q += l_rn
        cig = ""
        for _ in range(n_cig):
            (c,) = struct.unpack_from("<I", data, q)
            cig += f"{c >> 4}{'MIDNSHP=X'[c & 15]}"
            q += 4
        sb = data[q:q + (l_seq + 1) // 2]
        seq = "".join(code[b >> 4] + code[b & 15] for b in sb)[:l_seq]
        q += (l_seq + 1) // 2
        qual = data[q:q + l_seq]
        recs.append({"name": name, "ref": refs[ref_id][0], "ref_id": ref_id, "pos": pos, "mapq": mapq, "bin": bin_, "flag": flag,
                     "cigar": cig, "seq": seq, "qual": qual, "next_ref": nref, "next_pos": npos, "tlen": tlen})
        p += 4 + bs
    return text, refs, recs
