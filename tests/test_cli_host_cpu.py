"""CPU: the product's whole host program (gr_path_main in libgrpath_host.so: options, FASTQ
reader, read filters, Phred median, the three passes, the order-exact classifier, output
files) over an engine function table backed by the oracle — byte-identical files against
the oracle CLI, without a GPU.  (tests/test_gpu_cli.py does the same with the HIP engine.)"""
import ctypes as C
import filecmp
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

RUNNER = textwrap.dedent("""
    import ctypes as C, os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
    import orc
    from goldrush_amd import host
    from oracle_engine import OracleCliEngine
    eng = OracleCliEngine(orc, ingest=bool(os.environ.get("ORACLE_ENGINE_INGEST")))
    args = [b"goldrush_path"] + [a.encode() for a in sys.argv[1:]]
    arr = (C.c_char_p * (len(args) + 1))(*args, None)
    rc = host.load().gr_path_main(len(args), arr, C.byref(eng.vt))
    sys.stdout.flush(); sys.stderr.flush()
    if os.environ.get("ORACLE_ENGINE_INGEST"):  # every pin answered by an unpin before the next one / the end
        ops = [p[0] for p in eng.pins]
        assert eng.n_parse > 0, "the host did not take its chunked source"
        if os.environ.get("EXPECT_PINS"):
            assert "pin" in ops
            # the bodies of coming chunks were handed over ahead of their parse and every one of them was parsed with its
            # bytes unchanged (the oracle engine checks the contract of grp_fastq_prefetch: tests/oracle_engine.py)
            ps = eng.prefetch_stats
            assert ps["matched"] > 0 and ps["matched"] + ps["dropped"] <= ps["issued"] <= ps["matched"] + ps["dropped"] + 2, ps
        assert ops.count("pin") <= ops.count("unpin") and all(a != b for a, b in zip(ops, ops[1:]) if a == "pin"), ops
    if os.environ.get("GRP_TEST_MARK"):  # what the engine was asked to do (the ranks other than 0 must stay silent)
        open(os.environ["GRP_TEST_MARK"] + "." + os.environ.get("GRP_RANK", "0"), "w").write(str(getattr(eng, "n_bv_exports", 0)))
    os._exit(rc)
""")


@pytest.mark.parametrize("mode,batch", [("silver", 0), ("silver", 1), ("silver", 7), ("golden", 0), ("ntcard", 0), ("ntcard", 5), ("gz", 0), ("gz", 3)])
def test_host_program_over_oracle_engine_matches_oracle_cli(oracle, native, tmp_path, mode, batch):
    fq = os.path.join(GOLD, "tiny.fq")
    common = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j2", "-d5", "-x10", "-s1011011110110111101101", "-g60000", "-b4", "-H600000", "-i", fq, "--verbose"]
    if mode == "ntcard":  # the estimate replaces -H: drop it (and add a record with N, a short one, lower case)
        common = [a for a in common if not a.startswith("-H")] + ["--ntcard"]
        fq2 = str(tmp_path / "reads.fq")
        with open(fq, "rb") as src, open(fq2, "wb") as dst:
            dst.write(src.read())
            for i, seq in enumerate([b"ACGTACGTACGTACGTACGTACG", b"acgtnACGTACGTACGTACGTACGTACGTAC", b"ACGTAC" * 10 + b"RY" + b"TTGCA" * 9]):
                dst.write(b"@x%d\n%s\n+\n%s\n" % (i, seq, b"5" * len(seq)))
        common[common.index(fq)] = fq2
    args = common + (["-P0", "-r0.9", "--silver_path", "-M3", "-m1500"] if mode in ("silver", "gz") else ["-P12", "-m0"])
    args_p = list(args)
    if mode == "gz":  # gzip-compressed input for the product, the plain file for the oracle
        import gzip

        gz = str(tmp_path / "reads.fq.gz")
        with open(fq, "rb") as src, gzip.open(gz, "wb") as dst:
            dst.write(src.read())
        args_p[args_p.index(fq)] = gz
    d_o, d_p = tmp_path / "o", tmp_path / "p"
    d_o.mkdir()
    d_p.mkdir()
    ro = oracle.run_cli(args + ["-p", str(d_o / "out")], timeout=600)
    script = tmp_path / "runner.py"
    script.write_text(RUNNER.format(root=ROOT))
    rp = subprocess.run([sys.executable, str(script)] + args_p + ["-p", str(d_p / "out")], capture_output=True, text=True, timeout=900,
                        env=dict(os.environ, GRP_HOST_INGEST="1", OMP_NUM_THREADS="2", **({"GRP_BATCH_RECORDS": str(batch)} if batch else {})))
    assert rp.returncode == ro.returncode, (rp.returncode, ro.returncode, rp.stderr[-3000:])
    fo, fp = sorted(os.listdir(d_o)), sorted(os.listdir(d_p))
    assert fo == fp and fo, (fo, fp)
    for f in fo:
        assert filecmp.cmp(d_o / f, d_p / f, shallow=False), f
    assert any(os.path.getsize(d_p / f) > 0 for f in fo)
    keep = ("Expected entries", "Total expected entries", "\toccupancy", "Visited", "Saw:", "Assigned:", "Unassigned:", "Total queries", "Total hits", "Total misses", "Num reads", "m_filterSize", "\texpected hash space",
            "\tminimum average phred", "num_", "Total reads skipped")
    pick = lambda text: [l for l in text.splitlines() if l.startswith(keep)]  # noqa: E731
    assert pick(rp.stderr) == pick(ro.stderr)


@pytest.mark.parametrize("mode,chunk", [("silver", 0), ("silver", 30000), ("golden", 9000), ("golden", 100000), ("gz", 50000), ("silver", 4097)])
def test_chunked_source_with_reader_thread_matches_oracle_cli(oracle, native, tmp_path, mode, chunk):
    """The host's chunked FASTQ source (reader thread, two slots, the partial record carried into the
    next chunk, records longer than a chunk) over a Python restatement of the ingest entry points:
    chunk sizes from smaller than a record to larger than the file, plain and gzip input, the packed
    reads kept between the passes or parsed again."""
    fq = os.path.join(GOLD, "tiny.fq")
    common = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j2", "-d5", "-x10", "-s1011011110110111101101", "-g60000", "-b4", "-H600000", "-i", fq, "--verbose"]
    args = common + (["-P0", "-r0.9", "--silver_path", "-M3", "-m1500"] if mode in ("silver", "gz") else ["-P12", "-m0"])
    args_p = list(args)
    if mode == "gz":
        import gzip

        gz = str(tmp_path / "reads.fq.gz")
        with open(fq, "rb") as src, gzip.open(gz, "wb") as dst:
            dst.write(src.read())
        args_p[args_p.index(fq)] = gz
    d_o = tmp_path / "o"
    d_o.mkdir()
    ro = oracle.run_cli(args + ["-p", str(d_o / "out")], timeout=600)
    script = tmp_path / "runner.py"
    script.write_text(RUNNER.format(root=ROOT))
    for resident in ("on", "off"):
        d_p = tmp_path / ("p_" + resident)
        d_p.mkdir()
        env = dict(os.environ, ORACLE_ENGINE_INGEST="1", OMP_NUM_THREADS="2", GRP_RESIDENT=resident, **({"GRP_INGEST_CHUNK": str(chunk)} if chunk else {}))
        if chunk == 30000:  # the chunk buffer page-locked: every pin released before the next one and at the end
            env.update(GRP_PIN_MIN_BYTES="1", EXPECT_PINS="1")
        env.pop("GRP_HOST_INGEST", None)
        rp = subprocess.run([sys.executable, str(script)] + args_p + ["-p", str(d_p / "out")], capture_output=True, text=True, timeout=900, env=env)
        assert rp.returncode == ro.returncode, (rp.returncode, ro.returncode, rp.stderr[-3000:])
        fo, fp = sorted(os.listdir(d_o)), sorted(os.listdir(d_p))
        assert fo == fp and fo, (fo, fp)
        for f in fo:
            assert filecmp.cmp(d_o / f, d_p / f, shallow=False), (f, resident)
        keep = ("Visited", "Saw:", "Assigned:", "Unassigned:", "Total queries", "Total hits", "Total misses", "Num reads", "m_filterSize", "num_", "Total reads skipped")
        pick = lambda text: [l for l in text.splitlines() if l.startswith(keep)]  # noqa: E731
        assert pick(rp.stderr) == pick(ro.stderr)


@pytest.mark.parametrize("mode", ["silver", "golden", "golden_replicated_fill"])
def test_two_ranks_of_the_host_program_write_what_one_rank_writes(oracle, native, tmp_path, mode):
    """goldrush-path as two processes (GRP_WORLD=2: one per GPU on a real node; here two
    oracle-backed engines): every rank fills the bit vector from ITS share of the read batches and
    the vectors are OR-merged (round 3; staged through host memory here, RCCL inside the engine on a
    node with one GPU per rank), the ranks share the query work of every window and exchange the
    decisions through /dev/shm; rank 0 writes files identical to the oracle CLI's, rank 1
    writes nothing."""
    replicated = mode.endswith("_replicated_fill")
    mode = mode.split("_")[0]
    fq = os.path.join(GOLD, "tiny.fq")
    common = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j2", "-d5", "-x10", "-s1011011110110111101101", "-g60000", "-b4", "-H600000", "-i", fq, "--verbose"]
    args = common + (["-P0", "-r0.9", "--silver_path", "-M3", "-m1500"] if mode == "silver" else ["-P12", "-m0"])
    d_o, d_p = tmp_path / "o", tmp_path / "p"
    d_o.mkdir()
    d_p.mkdir()
    ro = oracle.run_cli(args + ["-p", str(d_o / "out")], timeout=600)
    script = tmp_path / "runner.py"
    script.write_text(RUNNER.format(root=ROOT))
    key = "test_cli_%d_%s" % (os.getpid(), mode)
    procs = []
    for rank in range(2):
        env = dict(os.environ, GRP_HOST_INGEST="1", OMP_NUM_THREADS="2", GRP_WORLD="2", GRP_RANK=str(rank), GRP_SHM_KEY=key, GRP_BATCH_RECORDS="9", GRP_TEST_MARK=str(tmp_path / "mark"))
        if replicated:
            env["GRP_REPLICATED_FILL"] = "1"
        procs.append(subprocess.Popen([sys.executable, str(script)] + args + ["-p", str(d_p / "out")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=900) for p in procs]
    assert [p.returncode for p in procs] == [ro.returncode, ro.returncode], [o[1][-2000:] for o in outs]
    fo, fp = sorted(os.listdir(d_o)), sorted(os.listdir(d_p))
    assert fo == fp and fo, (fo, fp)
    for f in fo:
        assert filecmp.cmp(d_o / f, d_p / f, shallow=False), f
    keep = ("Visited", "Saw:", "Assigned:", "Unassigned:", "Total queries", "Total hits", "Total misses", "Num reads", "m_filterSize")
    pick = lambda text: [l for l in text.splitlines() if l.startswith(keep)]  # noqa: E731
    assert pick(outs[0][1]) == pick(ro.stderr)
    assert outs[1][1].strip() == "" and outs[1][0].strip() == ""  # rank 1 is silent
    assert not os.path.exists("/dev/shm/grp_" + key)
    exports = [int(open(str(tmp_path / "mark") + "." + str(r)).read()) for r in range(2)]
    assert all(e == 0 for e in exports) if replicated else all(e >= 1 for e in exports)  # the sharded fill was merged


def test_a_garbage_line_ends_the_input_for_good_whatever_the_batching(oracle, native, tmp_path):
    """A line that is not a FASTQ header ends the input like an end of file — in every batching
    of the host reader (the GPU ingest latches the same state, tests/test_gpu_ingest.py): the
    run on the damaged file equals the run on the file cut in front of the damage."""
    lines = open(os.path.join(GOLD, "tiny.fq"), "rb").read().split(b"\n")
    n_ok = 20
    cut = tmp_path / "cut.fq"
    bad = tmp_path / "bad.fq"
    cut.write_bytes(b"\n".join(lines[: 4 * n_ok]) + b"\n")
    bad.write_bytes(b"\n".join(lines[: 4 * n_ok] + [b"garbage that is no header"] + lines[4 * n_ok:]))
    common = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j2", "-d5", "-x10", "-s1011011110110111101101", "-g60000", "-b4", "-H600000", "-P12", "-m0"]
    script = tmp_path / "runner.py"
    script.write_text(RUNNER.format(root=ROOT))
    outs = {}
    for name, path, batch in (("cut", cut, 0), ("bad", bad, 0), ("bad3", bad, 3), ("bad20", bad, 20), ("bad21", bad, 21)):
        d = tmp_path / name
        d.mkdir()
        rp = subprocess.run([sys.executable, str(script)] + common + ["-i", str(path), "-p", str(d / "out")], capture_output=True, text=True, timeout=900,
                            env=dict(os.environ, GRP_HOST_INGEST="1", OMP_NUM_THREADS="2", **({"GRP_BATCH_RECORDS": str(batch)} if batch else {})))
        assert rp.returncode == 0, rp.stderr[-2000:]
        outs[name] = open(d / "out.fa", "rb").read()
    assert outs["cut"] and all(v == outs["cut"] for v in outs.values())


@pytest.mark.parametrize("ingest", [False, True])
def test_a_truncated_gzip_file_is_an_error_not_a_shorter_input(oracle, native, tmp_path, ingest):
    """zlib hands out what it could inflate of a gzip file that was cut off and then reports the end of the
    data like at a proper end — the run must end with an error instead of classifying a prefix of the reads (host reader
    and the chunked source of the GPU ingest alike); the complete file passes."""
    import gzip

    raw = open(os.path.join(GOLD, "tiny.fq"), "rb").read()
    whole = tmp_path / "whole.fq.gz"
    cutoff = tmp_path / "cutoff.fq.gz"
    whole.write_bytes(gzip.compress(raw))
    z = whole.read_bytes()
    cutoff.write_bytes(z[: len(z) // 2])
    common = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j2", "-d5", "-x10", "-s1011011110110111101101", "-g60000", "-b4", "-H600000", "-P12", "-m0"]
    script = tmp_path / "runner.py"
    script.write_text(RUNNER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    if ingest:
        env.update(ORACLE_ENGINE_INGEST="1", GRP_INGEST_CHUNK="65536")
    else:
        env.update(GRP_HOST_INGEST="1")
    rcs = {}
    for name, path in (("whole", whole), ("cutoff", cutoff)):
        d = tmp_path / name
        d.mkdir()
        rp = subprocess.run([sys.executable, str(script)] + common + ["-i", str(path), "-p", str(d / "out")], capture_output=True, text=True, timeout=900, env=env)
        rcs[name] = (rp.returncode, rp.stderr)
    assert rcs["whole"][0] == 0, rcs["whole"][1][-2000:]
    assert rcs["cutoff"][0] != 0, "a truncated gzip file passed for a shorter input"
    assert "failed" in rcs["cutoff"][1] and "cutoff.fq.gz" in rcs["cutoff"][1], rcs["cutoff"][1][-2000:]


@pytest.mark.parametrize("mode", ["silver", "golden"])
def test_debug_output_matches_the_oracle(oracle, native, tmp_path, mode):
    """--debug (goldrush_path.cpp:60-70, 109-124, 267-273, 907-1086): the Phred lines of the fill
    pass, the median array, and per read the skipped records, name, tile count, the nine
    tile-state dumps of the smoothing passes, the assigned / unassigned counts and the
    decision — line by line equal to the oracle's."""
    fq = os.path.join(GOLD, "tiny.fq")
    common = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j2", "-d5", "-x10", "-s1011011110110111101101", "-g60000", "-b4", "-H600000", "-i", fq, "--debug"]
    args = common + (["-P0", "-r0.9", "--silver_path", "-M3", "-m3000"] if mode == "silver" else ["-P12", "-m2500"])
    d_o, d_p = tmp_path / "o", tmp_path / "p"
    d_o.mkdir()
    d_p.mkdir()
    ro = oracle.run_cli(args + ["-p", str(d_o / "out")], timeout=600)
    script = tmp_path / "runner.py"
    script.write_text(RUNNER.format(root=ROOT))
    rp = subprocess.run([sys.executable, str(script)] + args + ["-p", str(d_p / "out")], capture_output=True, text=True, timeout=900,
                        env=dict(os.environ, GRP_HOST_INGEST="1", OMP_NUM_THREADS="2", GRP_BATCH_RECORDS="11"))
    assert rp.returncode == ro.returncode, rp.stderr[-3000:]
    for f in sorted(os.listdir(d_o)):
        assert filecmp.cmp(d_o / f, d_p / f, shallow=False), f
    keep = ("name:", "num tiles:", "num assigned tiles:", "num unassigned tiles:", "unassigned", "complete assignment", "trimmed", "assigned", "too short", "skipping:",
            "hairpin or quality", "phred avg:", "phred delta:", "Number of reads used", "Median array:")
    pick = lambda text: [l for l in text.splitlines() if l.startswith(keep) or (l[:1].isdigit() and "\t" in l)]  # noqa: E731
    po, pp = pick(ro.stderr), pick(rp.stderr)
    assert len(po) > 100 and any(l.startswith("too short") for l in po) and any("\t" in l for l in po)
    assert pp == po
