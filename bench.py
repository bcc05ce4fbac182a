#!/usr/bin/env python3
"""bench.py — GoldRush-Path hot path (spaced-seed ntHash + miBF query / insert) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU with torch.distributed.run — and where nothing did
(`WORLD_SIZE` unset), `--gpus N` starts its N ranks itself (launch_ranks).  A launcher
whose WORLD_SIZE is not N is an error.  Rank 0 prints ONE JSON line.

Workload (default): BASELINE.json configs[2] "C2" — 10 M synthetic ONT-like reads,
mean 25 kb, G = 3e9, k=22 w=16 h=3 tile=1000, o=0.1 (m = 61 146 729 472 bits), golden
path mode, explicit -P.  `--config C1` is configs[1] (1 M reads, G = 100e6); N > 1 on
C2 is configs[3] "C3" (the same stream shared by the ranks: strong scaling).  The
reads are generated on the GPU (grp_synth_reads): all inputs are resident in HBM
before the timed region.

Untimed setup (reported under "aux"): bit-vector fill of all reads + rank build
(phase 1 of goldrush-path, goldrush_path.cpp:1199-1205).

Timed region = phase 2 over the WHOLE stream from read 0 on the freshly built miBF
(goldrush_path.cpp:1229-1256): the stream is cut into K steps of reads/K reads; a
step is the order-exact classification of its reads — every tile frame hashed and
probed (hash + miBF query), decisions taken in file order, accepted reads inserted
into the miBF *before* any later read is decided — bit-identical to the reference's
serial loop (goldrush_amd/csrc/host/gr_classifier.cpp).  The insert-heavy head of the
path is inside the timed region; "phases" reports the head (until the insert rate of
a 16 k-read slice drops below 1 %) and the steady state separately.
Warm-up steps run on a throw-away engine of a small geometry (kernel code objects,
LDS attributes, allocator): they never touch the measured stream.

metric  : reads/s through hash + miBF query (whole job, all ranks)
roofline: the query kernel.  `achieved` = ALGORITHMIC bytes (128 B per
          probe = two 64-B sectors in the reference's layout) / kernel time from HIP
          events around every launch on the library's own stream.  Beside it: the
          bytes that actually move (`moved_*`, from the committed rocprofv3 PMC pass:
          one 64-B bucket per probe) and the probe rate against the measured
          random-64-B-line ceiling of the memory system (profiles/*gather_ceiling.json).
roofline_insert / roofline_fill: k_batch_collect + apply (2 x 128 B per record slot) and k_fill
          (128 B per probe; `traffic` = its PMC read + write bytes).
cpu_baseline: the CPU oracle (restatement of the reference) on bounded samples of the
          same workload, rank 0, N=1 only.
aux     : end_to_end (fill + grp_finalize + the timed region: the whole goldrush-path run on resident
          reads), probe_accounting (executed against useful probes, the excess attributed), stream_keep
          (what the streaming windows kept across their in-launch inserts), phases.steady.fit (seconds
          per read and per insert over the steady-state slices), pipeline_shaped, oracle_check.
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PRESET = "1011011110110111101101"  # bin/goldrush:70
PHASE_SLICE = 16384  # reads per timing slice inside a step (head / steady-state split)

CONFIGS = {
    # BASELINE.json configs[1]
    "C1": {"reads": 1_000_000, "genome": 100e6, "h": 3, "name": "C1: 1M synthetic ONT reads, mean 25 kb, G=100e6"},
    # configs[2] (N = 1) / configs[3] (N = 8, the same stream)
    "C2": {"reads": 10_000_000, "genome": 3e9, "h": 3, "name": "C2: 10M synthetic ONT reads, mean 25 kb, G=3e9 (human-scale)"},
    # configs[4]: silver mode, 5 paths, 5 seed patterns; 312 GB of packed reads: streamed through HBM in batches
    "C4": {"reads": 50_000_000, "genome": 3e9, "h": 5, "silver": True, "max_paths": 5, "name": "C4: 50M synthetic ONT reads, mean 25 kb, G=3e9, h=5, M=5 silver paths"},
}


class ReadStream:
    """The synthetic read set as consecutive batches generated on the GPU (grp_synth_reads).
    One batch when the packed reads fit beside the miBF (everything resident before the timed
    region); otherwise batches of `batch` reads generated on demand, one resident at a time
    (C4: 50 M reads = 312 GB of 2-bit bases do not fit 288 GB of HBM)."""

    def __init__(self, native, n_reads: int, genome: int, batch: int, sigma: float = 0.25, repeat_frac: float = 0.0):
        self.native, self.n, self.G = native, n_reads, genome
        self.repeat_frac = repeat_frac  # share of the genome made of repeat families (grpath_synth.h); 0: the uniform genome
        self.plan = native.synth_read_plan(n_reads, genome, sigma=sigma)
        self.batch = batch if batch else n_reads
        self.n_batches = (n_reads + self.batch - 1) // self.batch
        self.cur = None  # (index, DeviceReads, ReadBatch, lens)
        self.eng = None
        self.synth_s = 0.0

    def bounds(self, b):
        return b * self.batch, min((b + 1) * self.batch, self.n)

    def get(self, b):
        """(first read, ReadBatch, lens) of batch b, resident in HBM"""
        if self.cur is None or self.cur[0] != b:
            self.drop()
            lo, hi = self.bounds(b)
            t0 = time.perf_counter()
            dr = self.native.synth_reads_range(self.plan, lo, hi, self.G, repeat_frac=self.repeat_frac)
            rb = self.eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
            self.synth_s += time.perf_counter() - t0
            self.cur = (b, dr, rb, np.ascontiguousarray(dr.lens, dtype=np.uint32))
        return self.bounds(b)[0], self.cur[2], self.cur[3]

    def pieces(self, first, count):
        """the range [first, first+count) of the stream as (ReadBatch, lens, first inside the batch, count) pieces"""
        pos, end = first, first + count
        while pos < end:
            b = pos // self.batch
            lo, rb, lens = self.get(b)
            hi = self.bounds(b)[1]
            n = min(end, hi) - pos
            yield rb, lens, pos - lo, n
            pos += n

    def download(self, first, count):
        out = []
        for rb, lens, lo, n in self.pieces(first, count):
            out += self.cur[1].download(lo, n)
        return out

    def drop(self):
        if self.cur is not None:
            self.eng.sync()
            self.cur[2].free()
            self.cur[1].free()
            self.cur = None


def _newest_profile(pattern: str):
    """newest profiles/<pattern> by the numbers in its name (r02_v10_... sorts after r02_v9_...)"""
    def _ver(path):
        return [int(x) for x in re.findall(r"\d+", os.path.basename(path))]

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=_ver)
    return files[-1] if files else None


WITNESS = 1024  # reads at either end of the stream whose GPU decisions are kept for the oracle's check


def cpu_baseline(dr, eng, cls, phases, silver, n_reads: int, m: int, pop: int, seeds, k: int, tile: int, genome: int, budget_s: float = 10.0, kept=None):
    """Like-for-like CPU baseline: the oracle's serial process_read loop (hash, query, decide,
    insert; OpenMP over tiles like the reference, goldrush_path.cpp:1229-1256) on the SAME
    filter as the measured run and on the same two regimes of the stream:
      head    reads 0.. of the stream on the freshly built miBF (the bit vector of ALL reads,
              exported from the GPU, empty ID arrays) — the insert-heavy start of the path;
      steady  the last reads of the stream on the GPU's END state (ID and count arrays
              exported from the GPU into the oracle's arrays).
    `value` = whole-stream estimate: the measured run's head / steady read counts at the two
    CPU rates."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc  # test infrastructure: used here only as the timed CPU baseline
    from goldrush_amd import host, synth

    orc.build()
    n_s = WITNESS
    head = dr.download(0, n_s)
    s0 = n_reads - n_s
    steady = dr.download(s0, n_s)
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"grp_bench_cpu_{os.getpid()}")
    os.makedirs(tmp, exist_ok=True)
    fq = os.path.join(tmp, "sample.fq")
    synth.write_fastq(fq, [(b"r%d" % i, s, b"5" * len(s)) for i, s in enumerate(head + steady)])
    del head, steady
    cores = int(host.load().gr_effective_cpus())  # affinity mask and cgroup quota
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        cpu_model = "unknown CPU"
    bits = eng.export_bits()  # the filter of the whole data set
    args = ["-k", str(k), "-w16", "-t", str(tile), "-h", str(len(seeds)), "-s", PRESET, "-g", str(genome), "-P10", "-d50", "-m0",
            "-j", str(cores), "-i", fq, "-p", os.path.join(tmp, "o")]
    p = orc.Path(args, external_bits=bits)
    assert p.ok and p.filter_size() == m
    del bits
    lib = orc.load()
    mh = p.mibf_handle()
    assert lib.orcpy_mibf_pop(mh) == pop, "the oracle's rank build disagrees with the GPU's"

    # The oracle replays exactly the reads whose GPU commits were kept (`kept`, in commit order): the head from read 0 on
    # the empty arrays — the same state the GPU path started from — and the last reads on the GPU's exported END state,
    # which is the state in front of a read only behind the last insert the GPU committed (nothing changes the filter
    # after that), and only as long as the oracle's replay has not inserted anything itself.
    # kind / tiles / assigned tiles / trim range / first ID (head) / hits / misses.
    by_read = {c[0]: c for c in (kept or [])}
    last_steady_insert = max([c[0] for c in (kept or []) if c[0] >= n_reads - n_s and c[1] in (2, 4)], default=-1)
    steady_oracle_inserts = [0]
    check = {"head_reads": 0, "steady_reads": 0, "identical": True, "first_difference": None,
             "what": "oracle process_read on the same filter vs the commits the GPU path kept for these reads: kind, tiles, assigned tiles, trim range, first ID, hits, misses"}
    ocounts = [None, None]  # the oracle's hits / misses counters before / after a read

    def compare(i, d, li0, li1):
        g = by_read.get(i if i < n_s else n_reads - 2 * n_s + i)  # sample index -> stream index
        if g is None or not check["identical"]:
            return
        if i >= n_s:
            steady_oracle_inserts[0] += 1 if d.decision in (2, 4) else 0
            if g[0] <= last_steady_insert or steady_oracle_inserts[0]:
                return  # the exported end state is not the state this read was decided against
        k4 = d.decision == 4
        mine = (d.decision, d.num_tiles, d.num_assigned, d.trim_start if k4 else 0, d.trim_end if k4 else 0, d.first_id if d.decision in (2, 4) else 0,
                li1["total_hits_per_path"] - li0["total_hits_per_path"], li1["total_misses_per_path"] - li0["total_misses_per_path"])
        gk4 = g[1] == 4
        theirs = (g[1], g[2], g[3], g[4] if gk4 else 0, g[5] if gk4 else 0, g[6] if g[1] in (2, 4) else 0, g[8], g[9])
        if i >= n_s:  # the oracle allocates IDs from the END state's counter: the first ID of an insert there is not the GPU's
            mine, theirs = mine[:5] + mine[6:], theirs[:5] + theirs[6:]
        check["head_reads" if i < n_s else "steady_reads"] += 1
        if mine != theirs:
            check["identical"] = False
            check["first_difference"] = {"read": g[0], "oracle": mine, "gpu": theirs}

    def timed(first, count, budget_s=budget_s):
        t0 = time.time()
        done = ins = 0
        t_check = 0.0
        for i in range(first, first + count):
            if kept is not None:
                tc = time.time()
                li0 = p.log_info()
                t_check += time.time() - tc
            d = p.process(i)
            done += 1
            ins += 1 if d.decision in (2, 4) else 0
            if kept is not None:
                tc = time.time()
                compare(i, d, li0, p.log_info())
                t_check += time.time() - tc
            if time.time() - t0 - t_check > budget_s:
                break
        return done, ins, time.time() - t0 - t_check

    # the reference's structure: 6 hashing threads ahead of the serial loop through an ordered queue
    # (read_hashing.cpp:77-117, goldrush_path.cpp:1219), OpenMP over the tiles of a read with -j threads
    producers = 6
    p.start_producers(producers, 0, n_s)
    h_done, h_ins, h_dt = timed(0, n_s, 4.0 * budget_s)  # the slow regime (~10-25 reads/s at C2) gets most of the ~50 s
    p.stop_producers()
    if silver:  # every silver path is an insert-heavy head: no steady state to compare
        p.close()
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)
        return {"value": h_done / h_dt, "unit": "reads/s", "cores": cores, "kind": "port", "oracle_check": check if kept is not None else None,
                "sample": f"same filter (bit vector of all {n_reads} reads exported from the GPU, m={m}); reads 0..{h_done} of the stream on the empty ID arrays ({h_ins} inserts): "
                          f"the regime of every silver path; oracle process_read loop (query + decide + insert) with {producers} hashing producers ahead of it, OpenMP over tiles, -j {cores}; {cpu_model}",
                "threads": f"{cores} OpenMP + {producers} hashing", "cpu_model": cpu_model}
    # the GPU's end state -> the oracle's arrays (chunks: the export stages through device memory)
    ids_p, cnt_p = lib.orcpy_mibf_data(mh), lib.orcpy_mibf_counts(mh)
    chunk = 1 << 28
    for a0 in range(0, pop, chunk):
        n = min(chunk, pop - a0)
        eng._check(eng.lib.grp_export_ids(eng._h, a0, n, C.c_void_p(ids_p + 4 * a0), C.c_void_p(cnt_p + 4 * a0)))
    st = cls.state()
    p.set_state(st["ids_inserted"], st["inserted_bases"], st["id"])
    p.start_producers(producers, n_s, n_s)
    s_done, s_ins, s_dt = timed(n_s, n_s)
    p.close()
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    h_rate, s_rate = h_done / h_dt, s_done / s_dt
    hr, sr = phases["head"]["reads"], phases["steady"]["reads"]
    whole = (hr + sr) / (hr / h_rate + sr / s_rate)
    return {"value": whole, "unit": "reads/s", "cores": cores, "kind": "port", "oracle_check": check if kept is not None else None,
            "sample": f"same filter (bit vector of all {n_reads} reads exported from the GPU, m={m}); head: reads 0..{h_done} on the empty ID arrays "
                      f"({h_ins} inserts); steady: reads {s0}..{s0 + s_done} on the GPU's end state (IDs / counts exported, {s_ins} inserts); oracle process_read loop "
                      f"(query + decide + insert) with {producers} hashing producers ahead of it through an ordered queue (the reference's structure), OpenMP over tiles, -j {cores}; {cpu_model}; "
                      f"value = the run's {hr} head / {sr} steady reads at the two CPU rates",
            "threads": f"{cores} OpenMP + {producers} hashing", "cpu_model": cpu_model,
            "head_reads_per_s": h_rate, "steady_reads_per_s": s_rate, "head_sample_reads": h_done, "steady_sample_reads": s_done}


def warm_up(native, host, steps: int, device: int, h: int):
    """W untimed steps on a throw-away engine (small geometry): loads every kernel of the
    path, sets the LDS attributes, warms the allocator.  Never touches the measured stream."""
    if steps <= 0:
        return
    k, w, tile = 22, 16, 1000
    G = 2_000_000
    hl = host.load()
    seeds = host.make_seed_pattern(PRESET, k, w, h)
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(w, G, h), 1, 0.1)
    per = 512
    dr = native.synth_reads(per * steps, G, mean_len=25000, min_len=20000)
    eng = native.Engine(k, h, tile, m, seeds, device=device)
    rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    eng.bv_insert(rb)
    eng.finalize()
    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=10, threshold=10, unassigned_min=5, assigned_max=1, k=k, h=h,
                          target_bases=int(0.9 * G), max_paths=1, silver_path=False, record=False)
    lens = np.ascontiguousarray(dr.lens, dtype=np.uint32)
    for i in range(steps):
        cls.run_range(rb._h, lens, i * per, per)
    eng.sync()
    cls.close()
    rb.free()
    eng.close()
    dr.free()


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: N fresh rank processes through torch.distributed.run (one per GPU,
    rendezvous on 127.0.0.1), this process only waits — it has not imported torch or touched HIP, and never does.
    Rank 0's JSON line goes to our stdout as it is (the children inherit it); the exit code is non-zero if any rank fails."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver: dmabuf IPC only (RCCL between processes)
    sys.stderr.write("bench: --gpus %d without a launcher: starting %d ranks (%s)\n" % (n, n, " ".join(cmd[1:9])))
    sys.stderr.flush()
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.stderr.write("bench: the %d-rank run failed (exit code %d)\n" % (n, rc))
    return rc if rc >= 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS), help="BASELINE.json workload (default C2; with --gpus 8 that is C3)")
    ap.add_argument("--reads", type=int, default=0, help="override the config's read count")
    ap.add_argument("--genome", type=float, default=0.0, help="override the config's genome size")
    ap.add_argument("--batch", type=int, default=0, help="reads per step for the whole job (default reads // steps: the K steps cover the stream exactly once)")
    ap.add_argument("--h", type=int, default=0)
    ap.add_argument("--silver", type=int, default=-1, help="silver-path mode with this many paths (-M); default: the config's")
    ap.add_argument("--stream-batch", type=int, default=-1, help="reads per resident batch (0: all resident; default: 2 M when the packed reads exceed 100 GB)")
    ap.add_argument("--max-window", type=int, default=0)
    ap.add_argument("--filter-scale", type=float, default=1.0, help="developer: the miBF's size x this (what -x / ntCard would have said for a larger genome): the same reads at another occupancy — "
                    "not the BASELINE workload, the line says so")
    ap.add_argument("--repeat-frac", type=float, default=0.0, help="share of the genome made of repeat families (2-6 kb units at 1-5 %% divergence, ~30 / ~1 000 / ~10 000 copies; include/grpath_synth.h); "
                    "0 = the headline workload's uniform genome, where two reads share k-mers only where they overlap.  Not the headline: the line's config says so and aux.repeats sums up what the speculation did")
    ap.add_argument("--len-sigma", type=float, default=0.25, help="sigma of the log-normal read lengths (mean 25 kb, floor 20 kb); 0.25 = the headline workload (no read above ~64 kb), "
                    "0.6 = a realistic ONT tail (reads of 100 kb and more: the decision path for reads of more than 64 tiles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline-shaped", action="store_true", help="skip aux.pipeline_shaped (the silver-mode pass over the same reads after the timed region)")
    ap.add_argument("--trace", action="store_true", help="per-slice timing / speculation statistics on stderr")
    ap.add_argument("--no-kernel-timing", action="store_true", help="developer: no HIP events around the launches (roofline fields become meaningless)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for single-GPU plumbing tests)")
    ap.add_argument("--verify-ranks", action="store_true", help="developer: N > 1, compare the classifier state of all ranks after the run")
    ap.add_argument("--share-gpu", action="store_true", help="developer: all ranks use GPU 0 (plumbing test on a 1-GPU box, use with --backend gloo)")
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            # no launcher around us: start the N ranks ourselves (the single command that uses the whole
            # node, like bin/goldrush:240-246's -j$t) — before anything in this process touches HIP
            raise SystemExit(launch_ranks(a.gpus))
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: the two must agree (the line's n_gpus is the number of ranks that ran)"
                         % (a.gpus, os.environ["WORLD_SIZE"]))

    cfg = dict(CONFIGS[a.config])
    if a.reads:
        cfg["reads"] = a.reads
    if a.genome:
        cfg["genome"] = a.genome
    if a.h:
        cfg["h"] = a.h
    if a.silver >= 0:
        cfg["silver"], cfg["max_paths"] = a.silver > 0, max(a.silver, 1)
    silver, max_paths = bool(cfg.get("silver")), int(cfg.get("max_paths", 1))
    n_reads = cfg["reads"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and "OMP_NUM_THREADS" not in os.environ:
        # the ranks share the node's cores (and its cgroup quota): every rank already keeps
        # one core busy spinning on its records, the OpenMP regions of the host get the rest
        cpus = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                cpus = min(cpus, max(1, int(quota) // int(period)))
        except Exception:
            pass
        os.environ["OMP_NUM_THREADS"] = str(max(1, cpus // world - 1))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if a.share_gpu:
        local_rank = 0
        # the ranks' persistent launches must all be resident on the one device together (their in-launch inserts wait
        # grid-wide): one workgroup per CU each, unless the caller says otherwise — a plumbing run, not a measurement
        os.environ.setdefault("GRP_STREAM_WGS_PER_CU", "1")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.path.exists("/sys/class/net/lo"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the hostname may not resolve
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)
    coll_dev = "cuda" if a.backend == "nccl" else "cpu"
    from goldrush_amd import host, native

    G, k, w, tile, h, block = int(cfg["genome"]), 22, 16, 1000, cfg["h"], 10
    hl = host.load()
    seeds = host.make_seed_pattern(PRESET, k, w, h)
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(w, G, h), 1, 0.1)
    if a.filter_scale != 1.0:
        m = int(m * a.filter_scale) // 64 * 64

    per_step = a.batch if a.batch else n_reads // a.steps
    if per_step < 1 or per_step * a.steps > n_reads:
        raise SystemExit("--reads %d too small for %d steps of %d reads (the stream is never wrapped)" % (n_reads, a.steps, per_step))

    # ---- warm-up: W steps on a throw-away engine ---------------------------------
    t0 = time.time()
    warm_up(native, host, a.warmup, local_rank, h)
    t_warm = time.time() - t0

    # ---- untimed setup: inputs resident in HBM, phase 1 -------------------------
    t0 = time.time()
    stream_batch = a.stream_batch if a.stream_batch >= 0 else (2_000_000 if n_reads * 6400 > 100e9 else 0)
    rs = ReadStream(native, n_reads, G, stream_batch, sigma=a.len_sigma, repeat_frac=a.repeat_frac)
    eng = native.Engine(k, h, tile, m, seeds, device=local_rank)
    # the -o the filter size was computed for: the engine allocates the phase-2 tables beside the fill (grp_set_occupancy_hint;
    # a hint — grp_finalize measures the occupancy as ever), what the goldrush-path binary does with its own -o
    if not a.share_gpu:  # (ranks sharing ONE device — the plumbing test of a one-GPU box — have no room for two sets of tables beside two bit vectors at C2's size)
        eng.set_occupancy_hint(0.1)
    rs.eng = eng
    rs.get(0)
    t_synth = time.time() - t0
    # N > 1: the ranks' node-local exchange (libgrpath_host's /dev/shm all-gather, C++: what the goldrush-path binary uses
    # between its ranks) — opened in front of the fill since round 5: the merge of the sharded fill goes through it too
    shm = None
    comm_aux = None
    if world > 1 and os.path.isdir("/dev/shm") and not os.environ.get("GRP_BENCH_NO_SHM"):
        key = "bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "run"))
        shm = hl.gr_shm_allgather_open(world, rank, key.encode(), 120.0)
    if world > 1:
        # the choice is collective: one rank without the exchange (a timeout, no /dev/shm in its
        # container) and EVERY rank takes the torch / gloo path — otherwise some ranks would enter
        # dist.new_group while the others skip it, and the stripe widths would differ per rank
        flag = torch.tensor([1 if shm else 0], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if shm:
                hl.gr_shm_allgather_close(shm)
            shm = None
            if rank == 0:
                sys.stderr.write("bench: no /dev/shm exchange on every rank, using torch.distributed for the merge and a gloo group for the records\n")
    t0 = time.time()
    if world == 1:
        for rb_, _, lo_, n_ in rs.pieces(0, n_reads):
            eng.bv_insert(rb_, lo_, n_)
        eng.sync()
    elif shm:
        # the fill shards by reads and its merge is a bitwise OR.  THE PRODUCT'S PATH (round 5; what the
        # goldrush-path binary does between its ranks, csrc/host/gr_ranks.cpp): the ranks agree on how to merge —
        # RCCL inside the engine (grp_comm_unique_id on rank 0, the id handed round through the shm exchange,
        # grp_comm_init, grp_bv_merge_ranks: ncclAllToAll of the slices, OR, ncclAllGather — 2 x (N-1)/N of one
        # vector per rank over xGMI), or, for ranks sharing one device (the plumbing runs of a 1-GPU box), staged
        # through host memory — torch.distributed is only the launcher here.
        vt = host.hip_engine_vt()
        plan = hl.gr_fill_merge_plan(C.byref(vt), eng._h, shm, world, rank, local_rank)
        if plan == 0:
            raise SystemExit("bench: the ranks cannot merge a sharded fill (no RCCL communicator and no staged form)")
        shard = (n_reads + world - 1) // world
        lo, hi = min(rank * shard, n_reads), min((rank + 1) * shard, n_reads)
        for rb_, _, lo_, n_ in rs.pieces(lo, hi - lo):
            eng.bv_insert(rb_, lo_, n_)
        eng.sync()
        mrc = hl.gr_fill_merge_run(C.byref(vt), eng._h, shm, world, rank, plan)
        if mrc != 0:
            raise SystemExit("bench: merging the ranks' bit vectors failed (%d): %s" % (mrc, (native.load().grp_last_error(eng._h) or b"").decode()))
        cw, cr, cv, cm = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_uint32()
        native.load().grp_comm_info(eng._h, C.byref(cw), C.byref(cr), C.byref(cv), C.byref(cm))
        comm_aux = {"merge": "rccl" if plan == 2 else "staged through host memory (ranks share a device, or no RCCL)", "world": int(cw.value) if plan == 2 else world,
                    "rccl_version": int(cv.value) or None, "merges_through_grp_bv_merge_ranks": int(cm.value),
                    "what": "csrc/host/gr_ranks.cpp (gr_fill_merge_plan / _run) over csrc/grp_comm.inc: the goldrush-path binary's own merge"}
    else:
        # no /dev/shm exchange: the same reduce-scatter + all-gather spelled out with torch.distributed (kept as the
        # fall-back; the product's path is the branch above)
        shard = (n_reads + world - 1) // world
        lo, hi = min(rank * shard, n_reads), min((rank + 1) * shard, n_reads)
        for rb_, _, lo_, n_ in rs.pieces(lo, hi - lo):
            eng.bv_insert(rb_, lo_, n_)
        eng.sync()
        n_words = eng.bv_words()
        slice_words = ((n_words + world - 1) // world + 3) // 4 * 4
        mine = torch.zeros(slice_words * world * 4, dtype=torch.uint8, device="cuda")
        # torch fills the buffer on ITS stream, the engine copies into it on its own (non-blocking)
        # stream: without this wait the zero fill can land behind the export and wipe bits
        torch.cuda.synchronize()
        eng.bv_export_device(mine.data_ptr())
        got = torch.empty_like(mine)  # slice `rank` of every rank's vector
        if coll_dev == "cuda":
            dist.all_to_all_single(got, mine)
        else:
            tmp = torch.empty(mine.numel(), dtype=torch.uint8)
            dist.all_to_all_single(tmp, mine.cpu())
            got.copy_(tmp)
        torch.cuda.synchronize()
        for p in range(1, world):
            eng.words_or_device(got.data_ptr(), got.data_ptr() + p * slice_words * 4, slice_words)
        merged = got[: slice_words * 4]
        if coll_dev == "cuda":
            dist.all_gather_into_tensor(mine, merged)
        else:
            tmp = torch.empty(mine.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(tmp, merged.cpu())
            mine.copy_(tmp)
        torch.cuda.synchronize()
        eng.bv_import_device(mine.data_ptr())
        del got, mine, merged
        comm_aux = {"merge": "torch.distributed (%s): no /dev/shm exchange between the ranks" % a.backend, "world": world, "rccl_version": None, "merges_through_grp_bv_merge_ranks": 0}
    t_fill = time.time() - t0
    fill_stats = eng.kernel_stats()["fill"]
    t0 = time.time()
    pop = eng.finalize()
    t_finalize = time.time() - t0
    finalize_parts = eng.finalize_times()
    if world > 1 and shm and hl.gr_ranks_same_u64(shm, world, pop) != 1:
        raise SystemExit("bench: the ranks' filters differ after the merge (rank %d: %d set bits)" % (rank, pop))

    # ---- phase 2: order-exact classification, windows sharded over the ranks ----
    allgather = None
    if world > 1:
        # The decisions the ranks exchange are tiny (32 B per read, a few KB per call) and
        # already sit in host memory: they go through shared memory (one node) or, failing
        # that, a CPU (gloo) group.  A GPU collective
        # here would need free compute units while the persistent query launch owns the
        # device, and two PCIe copies per call; RCCL is used where bulk data moves (the
        # bit-vector all-gather above).
        # (the /dev/shm exchange was opened in front of the fill: `shm`)
        if shm:
            # the exchange costs tens of microseconds here, not hundreds: shorter stripes
            # (less speculative work lost per insert) still hide it behind the launches
            os.environ.setdefault("GRP_STRIPE", str(64 + 8 * world))
        ctrl, ctrl_dev = None, "cpu"
        if not shm and a.backend == "nccl":
            try:
                ctrl = dist.new_group(backend="gloo")
            except Exception as e:  # no usable CPU transport: stay on RCCL, without persistent launches beside it
                if rank == 0:
                    sys.stderr.write("bench: no gloo group (%s); decisions go through RCCL, streaming windows off\n" % e)
                ctrl, ctrl_dev = None, "cuda"
                os.environ["GRP_STREAM"] = "off"
        bufs = {}  # per message size: in / out tensors (no allocation per call)

        def allgather(user, send, nbytes, recv):  # noqa: E306
            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
            dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
            b = bufs.get(nbytes)
            if b is None:
                b = (torch.empty(nbytes, dtype=torch.uint8, device=ctrl_dev), torch.empty(nbytes * world, dtype=torch.uint8, device=ctrl_dev))
                bufs[nbytes] = b
            if ctrl_dev == "cpu":
                b[0].numpy()[:] = src
                dist.all_gather_into_tensor(b[1], b[0], group=ctrl)
                dst[:] = b[1].numpy()
            else:
                b[0].copy_(torch.from_numpy(src))
                dist.all_gather_into_tensor(b[1], b[0])
                dst[:] = b[1].cpu().numpy()
            return 0

    cls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, threshold=10, unassigned_min=5, assigned_max=1, k=k, h=h,
                          target_bases=int(0.9 * G), max_paths=max_paths, silver_path=silver, max_window=a.max_window, world=world, rank=rank,
                          allgather=None if shm else allgather, record=False)
    if shm:
        hl.gr_classifier_set_allgather(cls._h, C.cast(hl.gr_shm_allgather, C.c_void_p), shm)
    # the oracle's witness (cpu_baseline leg): what the GPU path decided for the first and the last WITNESS reads of
    # the stream stays inside the classifier (no callback per read) and is compared with the oracle's serial loop
    witness = world == 1 and rs.n_batches == 1 and not a.no_cpu_baseline and n_reads >= 2 * WITNESS
    if witness:
        cls.keep_commits(0, WITNESS, n_reads - WITNESS, WITNESS)
    slices = []  # (reads, seconds, inserts) per timing slice, rank 0's clock
    finished = [False]  # silver mode: the last path is complete (the reference exits there)
    synth_before = rs.synth_s

    def step(i: int):
        # one step = the next per_step reads of the stream, run as slices of PHASE_SLICE reads
        # so that the head of the path and the steady state can be told apart
        first = i * per_step
        done = 0
        while done < per_step and not finished[0]:
            n = min(PHASE_SLICE, per_step - done)
            s0 = cls.state()
            ts = time.perf_counter()
            for rb_, lens_, lo_, n_ in rs.pieces(first + done, n):
                if not finished[0]:
                    finished[0] = cls.run_range(rb_._h, lens_, lo_, n_)
            te = time.perf_counter()
            s1 = cls.state()
            slices.append((s1["reads_committed"] - s0["reads_committed"], te - ts, s1["inserts"] - s0["inserts"]))
            if a.trace and rank == 0:
                sys.stderr.write("reads %d..%d: %.2f ms windows=%d queried=%d inserts=%d batches=%d undone=%d batch_reads=%d\n" % (
                    first + done, first + done + n, (te - ts) * 1e3, s1["windows"] - s0["windows"], s1["reads_queried"] - s0["reads_queried"], s1["inserts"] - s0["inserts"],
                    s1["batches"] - s0["batches"], s1["batches_undone"] - s0["batches_undone"], s1["batch_reads"] - s0["batch_reads"]))
            done += n

    if a.no_kernel_timing:
        eng.set_timing(False)
    eng.sync()
    eng.reset_kernel_stats()
    st0 = cls.state()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ks = eng.kernel_stats()
    st1 = cls.state()
    # Every read of the stream went into the bit vector, so every probe of every frame meets a set
    # bit and is counted as a hit or a miss (goldrush_path.cpp:567-594): anything else means the
    # filter lost bits on the way (a sharded fill merged wrongly, a fill that did not cover the reads).
    if eng.verify_stats()["impossible_deltas"]:
        raise SystemExit("bench: grp_batch_verify met %d tiles with an impossible delta (a logic error of the patch: the results stay exact, the run is refused)" % eng.verify_stats()["impossible_deltas"])
    if st1["hits"] + st1["misses"] != h * st1["queries"]:
        raise SystemExit("bench: hits + misses = %d, expected h x queries = %d: the filter does not hold every read of the stream"
                         % (st1["hits"] + st1["misses"], h * st1["queries"]))
    if world > 1 and a.verify_ranks:
        keys = ("valid_reads", "total_tiles", "assigned_tiles", "unassigned_tiles", "queries", "hits", "misses", "inserted_bases", "id", "ids_inserted",
                "reads_committed", "inserts")
        mine = [int(st1[k_]) for k_ in keys]
        allst = [None] * world
        dist.all_gather_object(allst, mine)
        assert all(x == allst[0] for x in allst), "ranks disagree: %r" % (allst,)
        if rank == 0:
            sys.stderr.write("verify-ranks: %d ranks agree on %s\n" % (world, dict(zip(keys, mine))))

    if rank == 0:
        kq = ks["query"]
        kernel_name = "k_query, throughput forms (streaming windows k_query<h,1,0,true,false>; large windows and the first query of a batch k_query<h,F,0,false,false>; the second query of a batch, through the batch's records, k_query<h,F,0,false,true>; F = 2 frames per lane up to h = 3, 1 beyond); the few latency windows are in aux.query_latency"
        if kq["units"] == 0:  # a run that never left the insert-heavy head: the latency form is all there is
            kq = ks["query_latency"]
            kernel_name = "k_query, windows of a few reads (k_query<h,F,0,false>; the run never left the insert-heavy head)"
        reads_done = st1["reads_committed"] - st0["reads_committed"]  # silver mode stops behind the last path
        avg_ms = kq["ms"] / max(kq["launches"], 1)
        probes_per_launch = kq["units"] / max(kq["launches"], 1)
        kq_s = kq["ms"] * 1e-3
        achieved = kq["units"] * 128 / kq_s / 1e9 if kq_s > 0 else 0.0
        gprobes = kq["units"] / kq_s / 1e9 if kq_s > 0 else 0.0
        # HBM bytes per launch: PMC counters cannot be read from inside this process; the
        # per-probe figure comes from the committed rocprofv3 --pmc pass of the same kernel
        # (newest profiles/r*_pmc_summary.json, tools/pmc_round.sh: TCC_EA0_RDREQ x 64 B = FETCH_SIZE x 1024 B)
        # Only a summary taken on a build of exactly these engine sources counts (the summary stores
        # the sha256 of goldrush_amd/csrc, tools/pmc_summary.py): a kernel change without a new PMC
        # pass reports traffic: null instead of silently keeping the old bytes per probe.
        traffic = bytes_per_probe_moved = fill_bytes_moved = None
        pmc_file = _newest_profile("r*_pmc_summary.json")
        pmc_note = "no PMC summary under profiles/"
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import importlib.util
            spec = importlib.util.spec_from_file_location("_pmc_hash", os.path.join(ROOT, "tools", "csrc_hash.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            pmc = json.load(open(pmc_file))
            if pmc.get("csrc_tree_sha256") == mod.csrc_tree_hash():
                fill_bytes_moved = (pmc.get("k_fill") or {}).get("hbm_bytes_per_probe")
                key = "k_query_all_variants" if "k_query_all_variants" in pmc else "k_query<3, 2, 0>"
                bytes_per_probe_moved = pmc[key]["hbm_bytes_per_probe"]
                traffic = bytes_per_probe_moved * probes_per_launch
                pmc_note = "rocprofv3 PMC passes of this build (%s), scaled by probes" % os.path.basename(pmc_file)
            else:
                pmc_note = "stale: %s was taken on other engine sources (csrc tree hash differs); run tools/pmc_round.sh" % os.path.basename(pmc_file)
        except Exception as e:
            pmc_note = "no usable PMC summary (%s)" % e
        # the measured random-64-B-line ceiling of this memory system (tools/gather_bench.hip,
        # archived by tools/profile_round.sh); null until a profile round has stored it
        ceiling = ceiling_file = None
        try:
            ceiling_file = _newest_profile("r*_gather_ceiling.json")
            ceiling = json.load(open(ceiling_file))["quad_64B_lines_Gps"]
        except Exception:
            pass
        # head of the path: slices up to the first one whose insert rate is below 1 %
        head_n = len(slices)
        for i, (n, _, ins) in enumerate(slices):
            if ins < 0.01 * n:
                head_n = i
                break
        head_reads = sum(s[0] for s in slices[:head_n])
        head_s = sum(s[1] for s in slices[:head_n])
        head_ins = sum(s[2] for s in slices[:head_n])
        tail_reads = sum(s[0] for s in slices[head_n:])
        tail_s = sum(s[1] for s in slices[head_n:])
        tail_ins = sum(s[2] for s in slices[head_n:])
        # what an insert costs the steady state: least squares of the slices' seconds on (reads, inserts) — the device
        # time the stream loses per insert, not the host's wait for the first record behind one (tools/scale_model.py)
        steady_fit = None
        if len(slices) - head_n >= 8 and tail_ins > 0:
            a_ = np.array([[s_[0], s_[2]] for s_ in slices[head_n:]], dtype=np.float64)
            b_ = np.array([s_[1] for s_ in slices[head_n:]], dtype=np.float64)
            sol_ = np.linalg.lstsq(a_, b_, rcond=None)[0]
            steady_fit = {"s_per_read": float(sol_[0]), "s_per_insert": float(sol_[1]), "slices": int(len(b_)),
                          "what": "least squares of the steady-state slices' seconds on their reads and inserts"}
        out = {
            "metric": "reads/s through GoldRush-Path (hash + miBF query)",
            "value": reads_done / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%s%s, k=22 w=16 h=%d tile=1000 o=0.1 -P10, %s, order-exact, whole stream from read 0 (insert-heavy head included)" % (
                           cfg["name"], " [C3: stream shared by %d GPUs]" % world if (world > 1 and a.config == "C2") else "", h,
                           "silver-path mode, %d paths (the run ends behind the last path: reads_timed = reads consumed)" % max_paths if silver else "golden-path mode"),
                       "reads": n_reads, "reads_timed": reads_done, "reads_per_step": per_step, "genome": G, "filter_bits": m, "pop": pop,
                       **({"repeat_frac": a.repeat_frac, "NOT_THE_HEADLINE_WORKLOAD": "repeat-rich genome (--repeat-frac)"} if a.repeat_frac > 0 else {}),
                       **({"filter_scale": a.filter_scale, "occupancy": pop / m, "NOT_THE_BASELINE_FILTER_SIZE": "miBF of %g x the size -x gives (--filter-scale)" % a.filter_scale} if a.filter_scale != 1.0 else {"occupancy": pop / m}),
                       "read_lengths": {"sigma": a.len_sigma, "mean": float(rs.plan[1].mean()), "max": int(rs.plan[1].max()), "reads_over_64_tiles": int((rs.plan[1] >= 65 * tile).sum())},
                       "parallelism": ("one GPU: windows committed as batches where >= ~1 % of the reads insert, streaming windows elsewhere" if world == 1 else "replicated miBF on %d GPUs: batches on every rank where >= ~1 %% of the reads insert, streaming windows striped over the ranks elsewhere (32-B decisions all-gathered per stripe group)" % world)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch: " + pmc_note,
                         "kernel": kernel_name,
                         "launches": kq["launches"], "avg_launch_ms": avg_ms,
                         "probes_per_launch": probes_per_launch, "bytes_per_probe": 128,
                         "note": "achieved/frac use the ALGORITHMIC 128 B per probe of SURVEY 8(d) (the reference's two-sector probe); this layout moves one 64-B bucket per probe, see moved_*; the kernel's real bound is the random-line rate, see line_rate_*",
                         "moved_bytes_per_probe": bytes_per_probe_moved,
                         "moved_GBps": (bytes_per_probe_moved * gprobes if bytes_per_probe_moved else None),
                         "moved_frac_of_peak": (bytes_per_probe_moved * gprobes / HBM_PEAK_GBPS if bytes_per_probe_moved else None),
                         "line_rate_Gprobes_per_s": gprobes,
                         "line_rate_ceiling_Gps": ceiling, "line_rate_ceiling_source": (os.path.basename(ceiling_file) if ceiling_file else None),
                         "line_rate_frac": (gprobes / ceiling if ceiling else None)},
            # the insert side (round 5, SURVEY 8(d): 2 x 128 B per unique rank — count RMW + ID RMW): k_batch_collect + k_batch_apply of the
            # batches in the timed region, per record slot ((frame, seed) of an inserted tile: what the pass walks; ~all of them unique ranks)
            "roofline_insert": ({"bound": "hbm", "kernel": "k_batch_collect + k_batch_apply (windows committed as batches)", "record_slots": int(ks["batch_insert"]["units"]), "kernel_ms": ks["batch_insert"]["ms"],
                                 "launches": int(ks["batch_insert"]["launches"]), "G_record_slots_per_s": ks["batch_insert"]["units"] / ks["batch_insert"]["ms"] / 1e6, "bytes_per_record_slot": 256,
                                 "achieved": ks["batch_insert"]["units"] * 256 / ks["batch_insert"]["ms"] / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": ks["batch_insert"]["units"] * 256 / ks["batch_insert"]["ms"] / 1e6 / HBM_PEAK_GBPS,
                                 "note": "the pass pays per memory request, not per byte (DESIGN 4): a record is a 64-B quad read of the query line, one compare-and-swap on the insert line of the same 128-B unit and a 4-B store"}
                                if ks["batch_insert"]["ms"] > 0 else None),
            # phase 1 (round 6; SURVEY 8(d): bit-vector fill = 128 B per probe, a 64-B sector read-modify-write): k_fill over every position
            # of every read, HIP events around its launches; untimed setup of the headline, but 40 % of a real run (aux.end_to_end)
            "roofline_fill": ({"bound": "hbm", "kernel": "k_fill<h> (grp_bv_insert: hash every read position, test-then-atomicOr of bit hash % m)", "probes": int(fill_stats["units"]),
                               "launches": int(fill_stats["launches"]), "kernel_ms": fill_stats["ms"], "avg_launch_ms": fill_stats["ms"] / max(fill_stats["launches"], 1),
                               "bytes_per_probe": 128, "achieved": fill_stats["units"] * 128 / fill_stats["ms"] / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": fill_stats["units"] * 128 / fill_stats["ms"] / 1e6 / HBM_PEAK_GBPS, "G_probes_per_s": fill_stats["units"] / fill_stats["ms"] / 1e6,
                               "traffic": (fill_bytes_moved * fill_stats["units"] / max(fill_stats["launches"], 1) if fill_bytes_moved else None),
                               "traffic_unit": "HBM bytes per launch (read + write): " + pmc_note, "moved_bytes_per_probe": fill_bytes_moved,
                               "line_rate_ceiling_Gps": ceiling, "line_rate_frac": (fill_stats["units"] / fill_stats["ms"] / 1e6 / ceiling if ceiling else None),
                               "note": "rank %d's share of the reads%s; a probe is one random 64-B line: the same request-rate wall as the query" % (rank, "" if world == 1 else " (1 / %d of the stream)" % world)}
                              if fill_stats["ms"] > 0 else None),
            "phases": {"slice_reads": PHASE_SLICE,
                       "head": {"reads": head_reads, "seconds": head_s, "reads_per_s": head_reads / head_s if head_s > 0 else None, "inserts": head_ins,
                                "definition": "slices before the first %d-read slice with an insert rate < 1 %%" % PHASE_SLICE},
                       "steady": {"reads": tail_reads, "seconds": tail_s, "reads_per_s": tail_reads / tail_s if tail_s > 0 else None, "inserts": tail_ins, "fit": steady_fit}},
            "aux": {"counters": {k_: int(st1[k_]) for k_ in ("valid_reads", "total_tiles", "assigned_tiles", "unassigned_tiles", "queries", "hits", "misses", "inserted_bases",
                                                             "ids_inserted", "reads_committed", "inserts")},  # the run's result: the same for every N, mode and switch
                    "pop": int(pop),
                    "comm": comm_aux,  # N > 1: which path merged the ranks' fills (rccl: grp_comm_* / grp_bv_merge_ranks inside the engine), the communicator's size, RCCL's version
                    # the whole goldrush-path run the reference times (goldrush_path.cpp:244,337,1208,1273): fill + setup + classification
                    "end_to_end": {"fill_s": t_fill, "finalize_s": t_finalize, "classify_s": dt, "total_s": t_fill + t_finalize + dt, "reads_per_s": reads_done / (t_fill + t_finalize + dt),
                                   "finalize_parts": finalize_parts,
                                   "what": "phase 1 (bit-vector fill of every read + rank build) + phase 2 (the timed region) over the same resident reads; FASTQ ingest excluded (tools/cli_end_to_end.py has the binary's)"},
                    "fill_probes": int(fill_stats["units"]),
                    "fill_reads_per_s": n_reads / t_fill, "fill_Gprobes_per_s": fill_stats["units"] * world / t_fill / 1e9, "fill_s": t_fill, "finalize_s": t_finalize,
                    "fill_mode": "single GPU" if world == 1 else "reads sharded over %d GPUs, bit vectors OR-merged as reduce-scatter (all-to-all + OR) + all-gather" % world, "synth_s": t_synth,
                    "read_batches": rs.n_batches, "synth_in_timed_region_s": rs.synth_s - synth_before,
                    "warmup_s": t_warm, "warmup_mode": "%d steps of 512 reads on a throw-away engine (G=2e6)" % a.warmup,
                    "timed": {key: st1[key] - st0[key] for key in ("windows", "reads_queried", "reads_committed", "inserts", "seconds_windows", "seconds_commit", "batches", "batches_undone", "batch_reads", "batches_fused", "stream_inserts", "stream_insert_fallbacks", "stream_relaunches", "stream_handbacks", "batch_overlap_cuts")},
                    "query_Gprobes_per_s": gprobes,
                    "query_kernel_s": kq_s, "decide_kernel_s": ks["decide"]["ms"] * 1e-3, "decide_launches": ks["decide"]["launches"],
                    "insert_kernel_s": ks["insert"]["ms"] * 1e-3, "insert_launches": ks["insert"]["launches"],
                    "query_latency": {"launches": ks["query_latency"]["launches"], "kernel_s": ks["query_latency"]["ms"] * 1e-3, "probes": ks["query_latency"]["units"],
                                      "what": "k_query<h,F,0,false> on windows of a few reads, summaries written straight to host memory"},
                    "kernel_stats": ks,  # every timed kernel family: launches, units, summed HIP-event ms (grpath.h GRP_K_*)
                    "stream_keep": eng.stream_stats(),  # round 6: what the in-launch inserts kept / redid of the tiles queried behind the inserting read
                    "batch_verify": eng.verify_stats(),  # tiles of the batches' second decisions: patched from records / queried again / redone / calls that fell back
                    "wall_s": dt},
        }
        # where the probes beyond the useful ones went: every frame of every read is useful once (h x queries);
        # the rest is speculation that was thrown away or confirmation that exactness asks for
        vs_, sk_ = out["aux"]["batch_verify"], out["aux"]["stream_keep"]
        per_tile = tile * h
        useful = h * int(st1["queries"] - st0["queries"])
        executed = int(ks["query"]["units"] + ks["query_latency"]["units"])
        acc = {
            "batches_second_query_of_tiles_without_records": vs_["queried"] * per_tile,       # the confirmation of the reads / tiles a batch did not insert
            "tiles_redone_with_the_worst_case_table_or_given_up_by_a_patch": (vs_["window_flagged"] + vs_["flagged"]) * per_tile,
            "streaming_tiles_queried_again_behind_an_insert": (sk_["tiles_redone_dirty"] + sk_["tiles_redone_lost"]) * per_tile,
        }
        acc["first_queries_thrown_away_and_windows_abandoned"] = max(executed - useful - sum(acc.values()), 0)  # batches taken back, the reads queried ahead of a batch that ended early, pipelined windows abandoned at an insert
        out["aux"]["probe_accounting"] = {"useful": useful, "executed": executed, "executed_over_useful": executed / useful if useful else None, "beyond_useful": acc,
                                          "share_of_useful": {k_: v_ / useful for k_, v_ in acc.items()} if useful else None,
                                          "note": "tile counts x tile x h (a read's clipped last tile counted whole); streaming: grp_debug_stream_stats — tiles finished in front of an in-launch insert whose probes met a changed slot or whose fingerprints were gone"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rs, eng, cls, out["phases"], silver, n_reads, m, pop, seeds, k, tile, G, kept=cls.kept_commits() if witness else None)
            check = out["cpu_baseline"].pop("oracle_check", None)
            out["aux"]["oracle_check"] = check
            if check and not check["identical"]:
                print(json.dumps(out), flush=True)
                raise SystemExit("bench: the GPU path's decisions differ from the oracle's serial loop: %s" % check["first_difference"])
        if world == 1 and not silver and not a.no_pipeline_shaped:
            # What bin/goldrush runs FIRST on the raw reads (bin/goldrush:253-260): --silver_path -M 5
            # -r 0.9 — every silver path starts on empty ID arrays (goldrush_path.cpp:156-187), so the
            # whole pass is the insert-heavy head regime, five times.  Same reads, same filter, IDs
            # reset; not the headline (BASELINE's metric is the golden-path stream above).
            eng.reset_ids()
            eng.sync()
            eng.reset_kernel_stats()
            scls = host.Classifier(eng._h, host.hip_engine_vt(), tile=tile, block=block, threshold=10, unassigned_min=5, assigned_max=1, k=k, h=h,
                                   target_bases=int(0.9 * G), max_paths=5, silver_path=True, max_window=a.max_window, record=False)
            t_s0 = time.perf_counter()
            s_fin, pos = False, 0
            while pos < n_reads and not s_fin:
                for rb_, lens_, lo_, n_ in rs.pieces(pos, min(PHASE_SLICE * 16, n_reads - pos)):
                    if not s_fin:
                        s_fin = scls.run_range(rb_._h, lens_, lo_, n_)
                pos += PHASE_SLICE * 16
            eng.sync()
            t_s = time.perf_counter() - t_s0
            sst = scls.state()
            if sst["hits"] + sst["misses"] != h * sst["queries"]:
                raise SystemExit("bench: silver pass: hits + misses != h x queries")
            out["aux"]["pipeline_shaped"] = {
                "what": "the pipeline's first goldrush-path process on these reads: --silver_path -M 5 -r 0.9 (bin/goldrush:253-260), same filter, ID arrays reset; the run ends behind the fifth path",
                "reads_consumed": int(sst["reads_committed"]), "seconds": t_s, "reads_per_s": sst["reads_committed"] / t_s if t_s > 0 else None,
                "inserts": int(sst["inserts"]), "paths_completed": int(sst["curr_path"]) - 1 if s_fin else int(sst["curr_path"]) - 1, "finished": bool(s_fin),
                "batches": int(sst["batches"]), "batches_undone": int(sst["batches_undone"]), "reads_queried": int(sst["reads_queried"]),
                "stream_inserts": int(sst["stream_inserts"]), "stream_rollovers": int(sst["stream_rollovers"]), "batch_overlap_cuts": int(sst["batch_overlap_cuts"]),
                "batches_refused": int(sst["batches_refused"]), "stream_handbacks": int(sst["stream_handbacks"]),
                "kernel_stats": {k_: v_ for k_, v_ in eng.kernel_stats().items() if v_["launches"]}}  # of this pass alone (HIP events, summed ms)
            del scls
        if a.repeat_frac > 0:
            # what the speculation did on a genome with repeats: exactness never depends on it, speed does
            ps = out["aux"].get("pipeline_shaped") or {}
            out["aux"]["repeats"] = {
                "repeat_frac": a.repeat_frac, "genome": "slots of 6144 bases, a repeat copy with probability 1.5 f: units of 2-6 kb at 1-5 % divergence in families of ~30 / ~1 000 / ~10 000 copies (include/grpath_synth.h)",
                "golden": {"reads_per_s": out["value"], "head_s": out["phases"]["head"]["seconds"], "steady_reads_per_s": out["phases"]["steady"]["reads_per_s"], "inserts": int(st1["inserts"] - st0["inserts"]),
                           "batches": int(st1["batches"] - st0["batches"]), "batches_taken_back": int(st1["batches_undone"] - st0["batches_undone"]),
                           "batches_refused_chain_overflow_or_size": int(st1["batches_refused"] - st0["batches_refused"]), "records_handed_back": int(st1["stream_handbacks"] - st0["stream_handbacks"]),
                           "verify": out["aux"]["batch_verify"]},
                "pipeline_shaped": {k_: ps.get(k_) for k_ in ("reads_per_s", "seconds", "inserts", "batches", "batches_undone", "batches_refused", "stream_handbacks")} if ps else None}
        print(json.dumps(out), flush=True)
    if world > 1:
        if shm:
            hl.gr_shm_allgather_close(shm)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
