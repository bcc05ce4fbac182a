"""GPU FASTQ ingest (include/grpath_ingest.h) against the host implementations:
record extents, id, ACGT flag, bit-identical Phred sums, 2-bit packing."""
import struct

import numpy as np
import pytest

from helpers import default_seeds, random_reads

pytestmark = pytest.mark.gpu


def _host_parse(text: bytes, final=True):
    """The host reader's semantics (gr_fastq.cpp): 4-line records, rtrim of CR / blank / tab,
    id up to the first whitespace; stops at a header that does not start with '@'."""
    lines, pos = [], 0
    while True:
        nl = text.find(b"\n", pos)
        if nl < 0:
            if final and pos < len(text):
                lines.append((pos, len(text)))
            break
        lines.append((pos, nl))
        pos = nl + 1
    recs = []
    for r in range(len(lines) // 4):
        ext = []
        for s, e in lines[4 * r: 4 * r + 4]:
            while e > s and text[e - 1:e] in (b"\r", b" ", b"\t"):
                e -= 1
            ext.append((s, e))
        (hs, he), (ss, se), _, (qs, qe) = ext
        if he == hs or text[hs:hs + 1] != b"@":
            return recs, True
        ie = hs + 1
        while ie < he and not text[ie:ie + 1].isspace():
            ie += 1
        recs.append((hs + 1, ie - hs - 1, ss, se - ss, qs, qe - qs))
    return recs, False


def _mk_text(seed, n=40, crlf=False, lower=False, with_n=False, no_final_nl=False):
    rng = np.random.default_rng(seed)
    out = []
    for i, s in enumerate(random_reads(n, 30, 3000, seed=seed)):
        if lower and i % 3 == 0:
            s = s.lower()
        if with_n and i % 5 == 2:
            s = s[:7] + b"N" + s[8:]
        q = bytes((rng.integers(0, 60, size=len(s)) + 33).astype(np.uint8))
        if i % 7 == 0:
            q = b"@" + q[1:]  # a quality line may start with '@'
        hdr = b"@read%d" % i + (b" extra\tcomment" if i % 4 == 0 else b"")
        eol = b"\r\n" if crlf else b"\n"
        out.append(hdr + eol + s + eol + b"+" + eol + q + eol)
    text = b"".join(out)
    if no_final_nl:
        text = text[:-(2 if crlf else 1)]
    return text


@pytest.mark.parametrize("kw", [dict(), dict(crlf=True), dict(lower=True, with_n=True), dict(no_final_nl=True)])
def test_parse_matches_host(native, kw):
    from goldrush_amd import host

    hl = host.load()
    eng = native.Engine(22, 3, 1000, 1 << 20, default_seeds(3))
    text = _mk_text(5, **kw)
    fq, rec, used, stopped = eng.fastq_parse(text)
    exp, _ = _host_parse(text)
    assert not stopped and used == len(text) and len(rec) == len(exp)
    for r, e in zip(rec, exp):
        assert (int(r["id_off"]), int(r["id_len"]), int(r["seq_off"]), int(r["seq_len"]), int(r["qual_off"]), int(r["qual_len"])) == e
        seq = text[e[2]: e[2] + e[3]]
        assert bool(r["flags"] & 1) == any(ch not in b"ACGTacgt" for ch in seq)
        qual = text[e[4]: e[4] + e[5]]
        # bit-identical left-to-right double sums (calc_phred_average.cpp:15-30)
        assert struct.pack("<d", float(r["phred_sum"])) == struct.pack("<d", hl.gr_sum_phred(qual, len(qual)))
        half = qual[: len(qual) // 2]
        exp_first = hl.gr_sum_phred(half, len(half)) if len(qual) >= 2 else 0.0
        assert struct.pack("<d", float(r["phred_first"])) == struct.pack("<d", exp_first)
    # packing of the ACGT records == host packing
    sel = [i for i, r in enumerate(rec) if not (r["flags"] & 1)]
    batch = eng.fastq_pack(fq, sel, rec["seq_len"][sel])
    seqs = [text[exp[i][2]: exp[i][2] + exp[i][3]] for i in sel]
    ref = eng.upload(seqs)
    for j in range(len(sel)):
        for t in range(len(seqs[j]) // 1000):
            assert np.array_equal(eng.tile_hashes(batch, j, t), eng.tile_hashes(ref, j, t))
    with pytest.raises(native.GrpError):
        eng.fastq_pack(fq, [i for i, r in enumerate(rec) if r["flags"] & 1][:1] or [10**6], [1])
    eng.fastq_free(fq)


def test_chunked_and_malformed(native):
    eng = native.Engine(22, 3, 1000, 1 << 20, default_seeds(3))
    text = _mk_text(9, n=25)
    exp, _ = _host_parse(text)
    # a chunk that ends in the middle of a record: only complete records are consumed
    cut = exp[10][4] + 5  # inside the quality line of record 10
    fq, rec, used, stopped = eng.fastq_parse(text[:cut], final_chunk=False)
    assert len(rec) == 10 and used == exp[10][0] - 1 and not stopped
    eng.fastq_free(fq)
    fq, rec2, used2, _ = eng.fastq_parse(text[used:], final_chunk=True)
    assert len(rec2) == 15 and used2 == len(text) - used
    eng.fastq_free(fq)
    # header not starting with '@' -> reader stops there
    bad = text[: exp[5][0] - 1] + b"X" + text[exp[5][0]:]
    fq, rec3, used3, stopped = eng.fastq_parse(bad)
    assert stopped and len(rec3) == 5
    eng.fastq_free(fq)
    # empty input, fewer than four lines
    fq, rec4, used4, stopped = eng.fastq_parse(b"")
    assert len(rec4) == 0 and used4 == 0
    eng.fastq_free(fq)
    fq, rec5, used5, _ = eng.fastq_parse(b"@a\nACGT\n+\n", final_chunk=True)
    assert len(rec5) == 0
    eng.fastq_free(fq)


@pytest.mark.parametrize("kw", [dict(), dict(lower=True, with_n=True), dict(no_final_nl=True)])
def test_ingest_matches_the_oracle_reader(oracle, native, tmp_path, kw):
    """The GPU ingest against the ORACLE's FASTQ reader (the restatement of btllib::SeqReader
    the oracle CLI reads its input with) and the oracle's Phred arithmetic: ids, sequences
    (case-folded), qualities, the ACGT filter and the raw Phred sums, record by record."""
    import ctypes as C

    text = _mk_text(21, n=60, **kw)
    path = str(tmp_path / "in.fq")
    with open(path, "wb") as f:
        f.write(text)
    lib = oracle.load()
    reads = oracle.orc_reads()
    assert lib.orc_reads_load(C.byref(reads), path.encode()) == 0 and reads.is_fastq
    eng = native.Engine(22, 3, 1000, 1 << 20, default_seeds(3))
    fq, rec, used, stopped = eng.fastq_parse(text)
    assert not stopped and used == len(text) and len(rec) == reads.n == 60
    for i, r in enumerate(rec):
        o = reads.rec[i]
        seq = text[int(r["seq_off"]): int(r["seq_off"]) + int(r["seq_len"])]
        qual = text[int(r["qual_off"]): int(r["qual_off"]) + int(r["qual_len"])]
        assert text[int(r["id_off"]): int(r["id_off"]) + int(r["id_len"])] == o.id
        assert seq.upper() == o.seq and len(seq) == o.len
        assert qual == o.qual and len(qual) == o.qlen
        assert bool(r["flags"] & 1) == (len(o.seq.strip(b"ACGT")) != 0)  # goldrush_path.cpp:293-301
        assert struct.pack("<d", float(r["phred_sum"])) == struct.pack("<d", lib.orc_sum_phred(o.qual, o.qlen))
    # the packed bases of the ACGT records hash like the oracle's strings
    sel = [i for i, r in enumerate(rec) if not (r["flags"] & 1) and r["seq_len"] >= 1000]
    batch = eng.fastq_pack(fq, sel, rec["seq_len"][sel])
    oseeds = oracle.Seeds(default_seeds(3))
    for j, i in enumerate(sel[:8]):
        for t in range(int(rec["seq_len"][i]) // 1000):
            assert np.array_equal(eng.tile_hashes(batch, j, t), oseeds.tile_hashes(reads.rec[i].seq, 1000, 22, t))
    eng.fastq_free(fq)
    lib.orc_reads_free(C.byref(reads))
    eng.close()


def test_pinned_chunk_buffer_over_several_passes(native):
    """grp_fastq_pin / _unpin: the same records from a page-locked chunk buffer, and — the fault this
    API replaced — a buffer that is freed and allocated again at (possibly) the same address between
    the passes is never taken for a registered one (the engine keeps no registration of its own)."""
    eng = native.Engine(22, 3, 1000, 1 << 20, default_seeds(3))
    text = _mk_text(21, n=900)
    fq, ref, used_ref, _ = eng.fastq_parse(text)
    eng.fastq_free(fq)
    for rounds in range(4):
        buf = np.empty(len(text) + (1 << 20), dtype=np.uint8)
        buf[: len(text)] = np.frombuffer(text, dtype=np.uint8)
        if rounds != 2:  # one pass on pageable memory in between
            eng.fastq_pin(buf)
        fq, rec, used, stopped = eng.fastq_parse_at(buf, len(text))
        assert used == used_ref and not stopped and np.array_equal(rec, ref)
        eng.fastq_free(fq)
        eng.fastq_unpin()  # before the buffer goes away (a second unpin is harmless)
        eng.fastq_unpin()
        del buf
    with pytest.raises(native.GrpError):
        eng.fastq_pin(np.zeros(0, dtype=np.uint8))


@pytest.mark.parametrize("shift", [0, 1, 5, 15, 16, 23])
def test_bodies_uploaded_ahead_of_their_parse(native, shift):
    """grp_fastq_prefetch (round 5): the bodies of the next two chunks are handed over before the chunk in front of them is
    parsed; a parse then gets [the unconsumed tail of the chunk before | body] and uploads the tail alone - the text begins
    wherever that leaves it inside a 16-byte group of the device buffer (`shift` moves the cuts, so the tail's length takes
    different values mod 16).  Same records, same bytes consumed, same packed bases as the parse of the same text from
    scratch; a third prefetch is refused; a parse of another text drops what is pending; (None, 0) forgets."""
    eng = native.Engine(22, 3, 1000, 1 << 20, default_seeds(3))
    text = _mk_text(31, n=1500)
    n = len(text)
    cuts = [0, n // 3 + shift, 2 * n // 3 + 3 * shift, n]
    room = 1 << 20
    bufs = []
    for a, b in zip(cuts, cuts[1:]):
        buf = np.zeros(room + (b - a), dtype=np.uint8)
        buf[room:] = np.frombuffer(text[a:b], dtype=np.uint8)
        bufs.append(buf)
    body = lambda k: (bufs[k][room:], bufs[k].size - room)  # noqa: E731

    # from scratch, chunk by chunk (no prefetch): the expectation
    exp, carry = [], b""
    for k in range(3):
        chunk = carry + bytes(bufs[k][room:])
        fq, rec, used, stopped = eng.fastq_parse(chunk, final_chunk=k == 2)
        sel = [j for j, r in enumerate(rec) if not (r["flags"] & 1) and r["seq_len"] >= 1000][:6]
        batch = eng.fastq_pack(fq, sel, rec["seq_len"][sel]) if sel else None
        hashes = [eng.tile_hashes(batch, j, 0).copy() for j in range(len(sel))]
        exp.append((rec.copy(), used, stopped, len(carry), sel, hashes))
        eng.fastq_free(fq)
        carry = chunk[used:]
    assert sum(len(e[0]) for e in exp) == 1500 and carry == b""
    assert shift == 0 or any(e[3] % 16 for e in exp[1:])

    # the host's order: chunk 0 is parsed, THEN the next two bodies go up; a third one has no buffer to go to
    fq, rec, used, stopped = eng.fastq_parse_at(bufs[0][room:], bufs[0].size - room, final_chunk=False)
    assert (used, stopped) == exp[0][1:3] and np.array_equal(rec, exp[0][0])
    assert eng.fastq_prefetch(*body(1)) and eng.fastq_prefetch(*body(2))
    assert not eng.fastq_prefetch(*body(0))
    eng.fastq_free(fq)
    carry = bytes(bufs[0][room:])[used:]
    for k in (1, 2):
        assert len(carry) == exp[k][3]
        bufs[k][room - len(carry): room] = np.frombuffer(carry, dtype=np.uint8)
        view = bufs[k][room - len(carry):]
        fq, rec, used, stopped = eng.fastq_parse_at(view, view.size, final_chunk=k == 2)
        assert (used, stopped) == exp[k][1:3] and np.array_equal(rec, exp[k][0]), k
        sel = exp[k][4]
        if sel:
            batch = eng.fastq_pack(fq, sel, rec["seq_len"][sel])
            for j in range(len(sel)):
                assert np.array_equal(eng.tile_hashes(batch, j, 0), exp[k][5][j]), (k, j)
        eng.fastq_free(fq)
        carry = bytes(view[used:])
    # a parse of another text drops what is pending (the body of chunk 1 again, then something else is parsed)
    assert eng.fastq_prefetch(*body(1))
    fq, rec, used, stopped = eng.fastq_parse(text[: cuts[1]], final_chunk=False)
    assert np.array_equal(rec, exp[0][0])
    eng.fastq_free(fq)
    assert eng.fastq_prefetch(*body(2)) and eng.fastq_prefetch(*body(1))  # two free buffers again: nothing was pending
    assert eng.fastq_prefetch(None, 0)  # forgotten: the buffers may go
    eng.close()
