"""GPU end-to-end: the goldrush-path CLI (C++ host + HIP engine) must write files
byte-identical to the oracle CLI (CPU restatement of the reference) on the same
seeded FASTQ: silver paths (process #1 of bin/goldrush:253-260) and the golden
path (process #2, bin/goldrush:240-248)."""
import filecmp
import glob
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk_fastq(path, genome_len, n_reads, mean_len, min_len, seed, noisy_qual=False, lower=False, with_n=0, short=0):
    from goldrush_amd import synth

    g = synth.random_genome(genome_len, seed)
    reads = synth.make_reads(g, n_reads, mean_len=mean_len, min_len=min_len, seed=seed + 1, noisy_qual=noisy_qual)
    rng = np.random.default_rng(seed + 2)
    out = []
    for i, (rid, seq, qual) in enumerate(reads):
        if lower and i % 7 == 0:
            seq = seq.lower()
        if with_n and i % with_n == 3:
            seq = seq[:100] + b"N" + seq[101:]
        if short and i % short == 1:
            seq, qual = seq[:800], qual[:800]
        if i % 11 == 5:  # low quality second half -> phred delta filter
            h = len(qual) // 2
            qual = qual[:h] + b"#" * (len(qual) - h)
        out.append((rid + (b" some comment" if i % 5 == 0 else b""), seq, qual))
    synth.write_fastq(path, out)
    return out


def _run_both(oracle, host, tmp_path, args, tag, env=None):
    d_o = tmp_path / f"{tag}_o"
    d_p = tmp_path / f"{tag}_p"
    d_o.mkdir()
    d_p.mkdir()
    ro = oracle.run_cli(args + ["-p", str(d_o / "out")], timeout=900)
    rp = subprocess.run([host.CLI_PATH] + args + ["-p", str(d_p / "out")], capture_output=True, text=True, timeout=900,
                        env=dict(os.environ, **(env or {})))
    assert rp.returncode == ro.returncode, (rp.returncode, ro.returncode, rp.stderr[-2000:])
    fo = sorted(os.path.basename(p) for p in glob.glob(str(d_o / "*")))
    fp = sorted(os.path.basename(p) for p in glob.glob(str(d_p / "*")))
    assert fo == fp, (fo, fp)
    for f in fo:
        assert filecmp.cmp(d_o / f, d_p / f, shallow=False), f"{f} differs"
    return ro, rp, d_o, d_p, fo


def _verbose_counters(stderr):
    keep = ("Visited", "Saw:", "Assigned:", "Unassigned:", "Total queries", "Total hits", "Total misses", "Num reads", "Average Phred",
            "m_filterSize", "expected hash space", "minimum average phred", "num_", "Total reads skipped")
    return [l for l in stderr.splitlines() if l.strip().startswith(keep)]


@pytest.fixture(scope="module")
def host(native):
    from goldrush_amd import host as h

    assert os.path.exists(h.CLI_PATH), "goldrush-path binary missing: run __graft_entry__.build()"
    return h


def test_silver_then_golden_byte_identical(oracle, host, tmp_path):
    fq = str(tmp_path / "reads.fq")
    # small genome, but the filter sized (-H) like a real run so that random
    # collisions stay below the -x threshold; tiles of 500, ID blocks of 4 tiles
    _mk_fastq(fq, 300_000, 900, 6000, 4000, seed=5, lower=True, with_n=50, short=40)
    common = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j8", "-d5", "-x10", "-s1011011110110111101101", "-g300000", "-b4", "-H4000000"]
    silver = common + ["-P0", "-r0.9", "--silver_path", "-M3", "-m3500", "-i", fq, "--verbose"]
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, silver, "silver")
    assert files == ["out_1.fq", "out_2.fq", "out_3.fq"]
    assert all(os.path.getsize(d_p / f) > 0 for f in files)
    text = b"".join(open(d_p / f, "rb").read() for f in files)
    assert text.count(b"_untrimmed\n") > 20 and text.count(b"_trimmed\n") > 20  # both insert kinds exercised
    assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)
    # process #2: cat the silver paths, golden-path mode, -m 0 (bin/goldrush:246-251)
    allfq = str(tmp_path / "all.fq")
    with open(allfq, "wb") as out:
        for f in files:
            out.write(open(d_p / f, "rb").read())
    golden = common + ["-P0", "-m0", "-i", allfq, "--verbose"]
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, golden, "golden")
    assert files == ["out.fa"] and os.path.getsize(d_p / "out.fa") > 0
    assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)


def test_designed_seed_h5_noisy_quals(oracle, host, tmp_path):
    """no -s preset (glibc-rand seed design), h=5, explicit -P, noisy qualities,
    fewer paths than requested (WARNING branch, normal end of file)."""
    fq = str(tmp_path / "reads.fq")
    _mk_fastq(fq, 150_000, 160, 8000, 6000, seed=9, noisy_qual=True)
    args = ["-k20", "-w14", "-t500", "-u5", "-a1", "-o0.1", "-h5", "-j2", "-P12", "-d5", "-x8", "-g150000", "-b4", "-r0.9", "--silver_path", "-M5", "-m5000",
            "-i", fq, "--verbose"]
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, args, "h5")
    assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)
    assert ("WARNING: Expected" in rp.stderr) == ("WARNING: Expected" in ro.stderr)


@pytest.mark.parametrize("k,w,h", [(40, 20, 3), (60, 24, 5)])
def test_spans_beyond_32_bases_through_the_binary(oracle, host, tmp_path, k, w, h):
    """-k 40 / -k 60 (round 4: k + h - 1 up to 64): designed seeds wider than one 64-bit window of 2-bit bases through the
    whole program — ingest, fill, silver paths — byte-identical to the oracle's."""
    fq = str(tmp_path / "reads.fq")
    _mk_fastq(fq, 150_000, 140, 8000, 6000, seed=12 + k)
    args = ["-k%d" % k, "-w%d" % w, "-t500", "-u5", "-a1", "-o0.1", "-h%d" % h, "-j2", "-P10", "-d5", "-x8", "-g150000", "-b4", "-r0.9", "--silver_path", "-M3", "-m5000",
            "-i", fq, "--verbose"]
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, args, "wide%d" % k)
    assert files and _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)


def _ntcard_lines(stderr):
    keep = ("Calculating expected entries", "Expected entries for seed pattern", "Total expected entries", "m_filterSize", "\texpected hash space",
            "\toccupancy")
    return [l for l in stderr.splitlines() if l.startswith(keep)]


@pytest.mark.parametrize("host_ingest", [False, True])
def test_ntcard_sizes_the_filter(oracle, host, tmp_path, host_ingest):
    """--ntcard (goldrush_path.cpp:1109-1112, ntcard.hpp): expected entries per seed from
    the reads themselves, incl. records with N, lower case and records shorter than the
    spans; the estimate sizes the filter, the rest of the run is unchanged."""
    fq = str(tmp_path / "reads.fq")
    recs = _mk_fastq(fq, 200_000, 400, 5000, 3000, seed=21, lower=True, with_n=9, short=40)
    with open(fq, "ab") as f:
        for i, seq in enumerate([b"ACGT" * 5 + b"AC", b"ACGT" * 5 + b"ACG", b"N" * 40, b"acgtnACGTACGTACGTACGTACGTACGTAC", b"ACGTAC" * 10 + b"RY" + b"TTGCA" * 9]):
            f.write(b"@extra%d\n%s\n+\n%s\n" % (i, seq, b"5" * len(seq)))
    args = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j4", "-d5", "-x10", "-s1011011110110111101101", "-g200000", "-b4", "-P10", "-m0",
            "--ntcard", "-i", fq, "--verbose"]
    env = {"GRP_HOST_INGEST": "1"} if host_ingest else {"GRP_INGEST_CHUNK": "300000"}
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, args, "ntc", env=env)
    assert files == ["out.fa"] and os.path.getsize(d_p / "out.fa") > 0
    lo, lp = _ntcard_lines(ro.stderr), _ntcard_lines(rp.stderr)
    assert lo == lp and len(lo) == 8, (lo, lp)
    assert "\toccupancy: 0.1000" in lp  # the reference's setprecision(4) << fixed stays on std::cerr
    assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)
    # -H given: --ntcard is ignored (goldrush_path.cpp:1109)
    ro, rp, _, _, _ = _run_both(oracle, host, tmp_path, args + ["-H2000000"], "ntc_H", env=env)
    assert "Calculating expected entries" not in rp.stderr and "Calculating expected entries" not in ro.stderr
    assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)


def test_c0_demo_configuration(oracle, host, tmp_path):
    """BASELINE configs[0] (the reference's own demo, tests/goldrush_test_demo.sh: G = 1e6,
    k22 w16 h3, tile 1000, -P 0, 5 silver paths of 0.9 G, then the golden path over them) on
    a synthetic 1 Mbp genome with 25 kb reads: both goldrush-path processes of bin/goldrush,
    byte-identical to the oracle CLI."""
    from goldrush_amd import synth

    fq = str(tmp_path / "test_reads.fq")
    synth.make_fastq(fq, 1_000_000, 1200, genome_seed=1, seed=2)
    common = ["-k22", "-w16", "-t1000", "-u5", "-a1", "-o0.1", "-h3", "-j4", "-d5", "-x10", "-s1011011110110111101101", "-g1000000", "-b10"]
    silver = common + ["-P0", "-r0.9", "--silver_path", "-M5", "-m20000", "-i", fq, "--verbose"]
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, silver, "c0_silver")
    assert files == ["out_%d.fq" % i for i in range(1, 6)] and all(os.path.getsize(d_p / f) > 0 for f in files)
    assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)
    allfq = str(tmp_path / "all.fq")
    with open(allfq, "wb") as out:
        for f in files:
            out.write(open(d_p / f, "rb").read())
    golden = common + ["-P0", "-m0", "-i", allfq, "--verbose"]
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, golden, "c0_golden")
    assert files == ["out.fa"] and os.path.getsize(d_p / "out.fa") > 0
    assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)


def test_gzip_input(oracle, host, tmp_path):
    """gzip-compressed FASTQ (GPU ingest of the inflated chunks, and the host reader): the
    same files as the oracle on the plain text."""
    import gzip

    fq = str(tmp_path / "reads.fq")
    _mk_fastq(fq, 150_000, 260, 6000, 4000, seed=31, lower=True, with_n=13)
    gz = fq + ".gz"
    with open(fq, "rb") as src, gzip.open(gz, "wb", compresslevel=1) as dst:
        dst.write(src.read())
    args = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j4", "-d5", "-x10", "-s1011011110110111101101", "-g150000", "-b4", "-H2500000", "-P0", "-r0.9",
            "--silver_path", "-M2", "-m3500", "--verbose"]
    d_o = tmp_path / "o"
    d_o.mkdir()
    ro = oracle.run_cli(args + ["-i", fq, "-p", str(d_o / "out")], timeout=900)
    for tag, env in (("gpu_ingest", {"GRP_INGEST_CHUNK": "400000"}), ("host_ingest", {"GRP_HOST_INGEST": "1"})):
        d_p = tmp_path / tag
        d_p.mkdir()
        rp = subprocess.run([host.CLI_PATH] + args + ["-i", gz, "-p", str(d_p / "out")], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert rp.returncode == ro.returncode, rp.stderr[-2000:]
        files = sorted(os.listdir(d_o))
        assert files == sorted(os.listdir(d_p)) and files
        for f in files:
            assert filecmp.cmp(d_o / f, d_p / f, shallow=False), (tag, f)
        assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)


def test_cli_error_paths(oracle, host, tmp_path):
    fa = tmp_path / "x.fa"
    fa.write_text(">r1\nACGT\n")
    base = ["-k22", "-w16", "-s1011011110110111101101", "-g1e5", "-P10"]
    for extra, code in [(["-i", str(fa)], 1),            # not FASTQ (goldrush_path.cpp:247-250)
                        ([], None)]:
        args = base + extra + ["-p", str(tmp_path / "o")]
        rp = subprocess.run([host.CLI_PATH] + args, capture_output=True, text=True, timeout=300)
        ro = oracle.run_cli(args, timeout=300)
        if code is not None:
            assert rp.returncode == ro.returncode == code
    # flag validation happens before any GPU work (opt.cpp:176-216)
    for bad in (["-w16", "-g1e5"], ["-k22", "-g1e5"], ["-k22", "-w16"], ["-k22", "-w16", "-g1e5", "-s101"], ["-k3", "-w3", "-g1e5", "-s101"]):
        rp = subprocess.run([host.CLI_PATH] + bad, capture_output=True, text=True, timeout=60)
        ro = oracle.run_cli(bad, timeout=60)
        assert rp.returncode == ro.returncode == 1
        assert rp.stdout == ro.stdout  # usage text
    rp = subprocess.run([host.CLI_PATH, "--help"], capture_output=True, text=True, timeout=60)
    ro = oracle.run_cli(["--help"], timeout=60)
    assert rp.returncode == ro.returncode == 0 and rp.stdout == ro.stdout
    # all reads filtered -> exit(1) (goldrush_path.cpp:327-334)
    fq = str(tmp_path / "short.fq")
    _mk_fastq(fq, 50_000, 20, 3000, 2000, seed=3)
    args = base + ["-i", fq, "-m20000", "-p", str(tmp_path / "o2")]
    rp = subprocess.run([host.CLI_PATH] + args, capture_output=True, text=True, timeout=300)
    ro = oracle.run_cli(args, timeout=300)
    assert rp.returncode == ro.returncode == 1


@pytest.mark.parametrize("env", [{"GRP_HOST_INGEST": "1", "GRP_BATCH_RECORDS": "1"}, {"GRP_HOST_INGEST": "1", "GRP_BATCH_RECORDS": "7"},
                                 {"GRP_HOST_INGEST": "1", "GRP_BATCH_RECORDS": "64"}, {"GRP_INGEST_CHUNK": "9000"}, {"GRP_INGEST_CHUNK": "70001"},
                                 {"GRP_INGEST_CHUNK": "1000000"}, {}])
def test_batch_boundaries(oracle, host, tmp_path, env):
    """Tiny host batches / tiny GPU-ingest chunks (smaller than one record: the buffer
    has to grow): the filter set, the skipped-read counter, the classifier state and
    the silver-path rollover all have to carry across uploads; host reader and GPU
    ingest must give the same files as the oracle."""
    fq = str(tmp_path / "reads.fq")
    _mk_fastq(fq, 200_000, 420, 6000, 4000, seed=15, lower=True, with_n=17, short=9)
    args = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j4", "-d5", "-x10", "-s1011011110110111101101", "-g200000", "-b4", "-H3000000",
            "-P0", "-r0.9", "--silver_path", "-M2", "-m3500", "-i", fq, "--verbose"]
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, args, "b", env=env)
    assert len(files) == 2
    assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)


def test_crlf_and_missing_final_newline(oracle, host, tmp_path):
    fq = tmp_path / "reads.fq"
    _mk_fastq(str(fq), 150_000, 150, 6000, 4000, seed=25)
    data = fq.read_bytes().replace(b"\n", b"\r\n")[:-2]
    fq.write_bytes(data)
    args = ["-k22", "-w16", "-t500", "-u5", "-a1", "-h3", "-j2", "-d5", "-x10", "-s1011011110110111101101", "-g150000", "-b4", "-H3000000",
            "-P0", "-r0.9", "--silver_path", "-M2", "-m3500", "-i", str(fq), "--verbose"]
    for env in ({}, {"GRP_HOST_INGEST": "1"}):
        ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, args, "crlf" + str(len(env)), env=env)
        assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)


def test_c4_geometry_on_a_subsample(oracle, host, tmp_path):
    """BASELINE configs[4]'s parameters — defaults (k=22 w=16 tile 1000, blocks of 10 tiles) with
    -h 5 seed patterns and -M 5 silver paths — on a sub-sample: 25 kb reads of a 1.2 Mbp genome,
    enough for all five paths (the run ends behind the fifth one, goldrush_path.cpp:173-176)."""
    fq = str(tmp_path / "reads.fq")
    _mk_fastq(fq, 1_200_000, 700, 25000, 20000, seed=31)
    args = ["-k22", "-w16", "-h5", "-s1011011110110111101101", "-g1200000", "-P10", "-j8", "--silver_path", "-M5", "-i", fq, "--verbose"]
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, args, "c4")
    assert files == ["out_%d.fq" % i for i in range(1, 6)] and all(os.path.getsize(d_p / f) > 0 for f in files)
    assert "WARNING: Expected" not in rp.stderr  # all five paths were generated
    assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)


def test_debug_dumps_match_the_oracle(oracle, host, tmp_path):
    """--debug through the HIP engine: the per-read lines and the nine tile-state dumps of the
    smoothing passes (goldrush_path.cpp:109-124, 907-1086) equal the oracle's, line by line."""
    fq = str(tmp_path / "reads.fq")
    _mk_fastq(fq, 200_000, 300, 6000, 4000, seed=41, short=9)
    args = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j4", "-d5", "-x10", "-s1011011110110111101101", "-g200000", "-b4", "-H3000000", "-P12", "-r0.9",
            "--silver_path", "-M2", "-m3500", "-i", fq, "--debug"]
    ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, args, "dbg")
    keep = ("name:", "num tiles:", "num assigned tiles:", "num unassigned tiles:", "unassigned", "complete assignment", "trimmed", "assigned", "too short", "skipping:",
            "hairpin or quality", "phred avg:", "phred delta:")
    pick = lambda text: [l for l in text.splitlines() if l.startswith(keep) or (l[:1].isdigit() and "\t" in l)]  # noqa: E731
    po, pp = pick(ro.stderr), pick(rp.stderr)
    assert len(po) > 1000 and any(l.startswith("trimmed") for l in po)
    assert pp == po


@pytest.mark.parametrize("env", [{}, {"GRP_RESIDENT": "off"}, {"GRP_RESIDENT_MAX_GB": "0.0002", "GRP_INGEST_CHUNK": "200000"}, {"GRP_INGEST_CHUNK": "30000"}])
def test_reads_kept_on_the_device_between_the_passes(oracle, host, tmp_path, env):
    """Round 3: the fill pass is the only full parse — its packed reads stay in HBM for the
    classification pass, the text of the reads written out comes back from the file by offset.  With a
    -f list (names compared through the file; one listed name is also a passing read's), a read whose
    name equals a failing read's (the reference filters by NAME), records in front of the first and
    behind the last classified read, a budget that runs out mid-file (fall back to the second parse),
    and the switch off: always the oracle's files and counters."""
    fq = tmp_path / "reads.fq"
    recs = _mk_fastq(str(fq), 200_000, 300, 6000, 4000, seed=35, lower=True, with_n=13, short=8)
    # a passing read that carries the name of a failing one (record 5 fails the delta filter)
    names = [r[0].split()[0] for r in recs]
    lines = fq.read_bytes().split(b"\n")
    lines[4 * 40] = b"@" + names[5]
    fq.write_bytes(b"\n".join(lines))
    flt = tmp_path / "skip.txt"
    flt.write_bytes(b"\n".join([names[2], names[100], names[101], names[299], b"not_a_read"]) + b"\n")
    base = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j4", "-d5", "-x10", "-s1011011110110111101101", "-g200000", "-b4", "-H3000000", "-P10", "-m3500", "-i", str(fq), "--verbose"]
    for tag, extra in (("silver", ["-r0.9", "--silver_path", "-M2"]), ("golden_f", ["-f", str(flt)])):
        ro, rp, d_o, d_p, files = _run_both(oracle, host, tmp_path, base + extra, tag + str(len(env)), env=env)
        assert _verbose_counters(rp.stderr) == _verbose_counters(ro.stderr)


def test_two_ranks_share_the_fill_and_the_windows(oracle, host, tmp_path):
    """goldrush-path as two processes on the one GPU of this box (GRP_WORLD=2, both on device 0):
    each rank fills the bit vector from its share of the read batches, the vectors are OR-merged
    (ranks sharing a device cannot form an RCCL communicator: the host-staged form through
    /dev/shm; on a node with a GPU per rank the same step is ncclAllToAll + OR + ncclAllGather
    inside the engine), the windows' query work is striped over the ranks.  Rank 0's files
    equal the oracle's; rank 1 writes nothing."""
    fq = str(tmp_path / "reads.fq")
    _mk_fastq(fq, 200_000, 420, 6000, 4000, seed=45, lower=True, with_n=17, short=9)
    for tag, extra in (("silver", ["-P0", "-r0.9", "--silver_path", "-M2", "-m3500"]), ("golden", ["-P10", "-m3500"])):
        args = ["-k22", "-w16", "-t500", "-u5", "-a1", "-o0.1", "-h3", "-j4", "-d5", "-x10", "-s1011011110110111101101", "-g200000", "-b4", "-H3000000", "-i", fq, "--verbose"] + extra
        d_o, d_p = tmp_path / (tag + "_o"), tmp_path / (tag + "_p")
        d_o.mkdir()
        d_p.mkdir()
        ro = oracle.run_cli(args + ["-p", str(d_o / "out")], timeout=900)
        key = "gpu_cli_%d_%s" % (os.getpid(), tag)
        procs = []
        for rank in range(2):
            env = dict(os.environ, GRP_WORLD="2", GRP_RANK=str(rank), GRP_LOCAL_RANK="0", GRP_SHM_KEY=key, GRP_STREAM_WGS_PER_CU="2", GRP_INGEST_CHUNK="400000")
            procs.append(subprocess.Popen([host.CLI_PATH] + args + ["-p", str(d_p / "out")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
        outs = [p.communicate(timeout=900) for p in procs]
        assert [p.returncode for p in procs] == [ro.returncode, ro.returncode], [o[1][-2000:] for o in outs]
        fo = sorted(os.path.basename(p) for p in glob.glob(str(d_o / "*")))
        fp = sorted(os.path.basename(p) for p in glob.glob(str(d_p / "*")))
        assert fo == fp and fo, (fo, fp)
        for f in fo:
            assert filecmp.cmp(d_o / f, d_p / f, shallow=False), f"{f} differs"
        assert _verbose_counters(outs[0][1]) == _verbose_counters(ro.stderr)
        assert outs[1][0].strip() == "" and outs[1][1].strip() == ""
