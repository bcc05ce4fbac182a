#!/usr/bin/env python3
"""bench.py — GoldRush-Path hot path (spaced-seed ntHash + miBF query / insert) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU with torch.distributed.run.  Rank 0 prints ONE
JSON line.

Workload (BASELINE.json configs[1], "C1"): 1 M synthetic ONT-like reads, mean
25 kb, G = 100e6, k=22 w=16 h=3 tile=1000, o=0.1.  The reads are generated on the
GPU (grp_synth_reads): all inputs are resident in HBM before the timed region.

Untimed setup (reported under "aux"): bit-vector fill of all reads + rank build
(phase 1 of goldrush-path, goldrush_path.cpp:1199-1205).

A step = the order-exact classification of the next `--batch` x N reads of the
stream (phase 2, goldrush_path.cpp:1229-1256, golden-path mode): every tile frame
is hashed and probed (hash + miBF query), the read decisions are taken in file
order and the accepted reads are inserted into the miBF *before* any later read
is decided — bit-identical to the reference's serial loop, with speculative GPU
windows in between (goldrush_amd/csrc/host/gr_classifier.cpp).  Warm-up steps
are the first W batches of the same stream (the insert-heavy start of the path).

metric  : reads/s through hash + miBF query (whole job, all ranks)
roofline: the query kernel, ALGORITHMIC bytes = 128 B per probe (two 64-B HBM
          sectors: bit+rank block, then ID), probes = frames x h; duration from
          HIP events recorded around every launch on the library's own stream.
cpu_baseline: the CPU oracle (restatement of the reference) on a bounded sample
          of the same workload, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PRESET = "1011011110110111101101"  # bin/goldrush:70


class ShmAllgather:
    """All-gather of small host buffers between the ranks of ONE node through a file in
    /dev/shm: every rank writes its block into the round's buffer, then publishes the round
    number; readers spin on the round numbers.  Two buffers alternate, so a rank that is
    one round ahead never overwrites what a slower rank still reads.  ~10 us per call
    against a few hundred for a loopback gloo ring: the 32-byte decision records are
    exchanged once per stripe group."""

    SLOT = 1 << 20  # bytes per rank and buffer

    def __init__(self, world: int, rank: int, key: str, barrier):
        import mmap

        self.world, self.rank, self.round = world, rank, 0
        self.path = "/dev/shm/grp_bench_%s" % key
        size = 4096 + 2 * world * self.SLOT
        if rank == 0:
            with open(self.path, "wb") as f:
                f.truncate(size)
        barrier()
        self._f = open(self.path, "r+b")
        self._mm = mmap.mmap(self._f.fileno(), size)
        buf = np.frombuffer(self._mm, dtype=np.uint8)
        self.seq = buf[:4096].view(np.uint64)[: world * 8 : 8]  # one 64-byte line per rank
        self.data = buf[4096:].reshape(2, world, self.SLOT)
        barrier()

    def __call__(self, src: np.ndarray, dst: np.ndarray):
        n = src.size
        assert n <= self.SLOT
        self.round += 1
        b = self.round & 1
        self.data[b, self.rank, :n] = src
        self.seq[self.rank] = self.round  # published after the data (x86 store order)
        for r in range(self.world):
            spins = 0
            while self.seq[r] < self.round:
                spins += 1
                if spins > 200000:
                    time.sleep(0)  # a rank that is far behind (start-up): do not burn its core
            dst[r * n:(r + 1) * n] = self.data[b, r, :n]

    def close(self, barrier):
        barrier()
        self._mm = None
        self._f.close()
        if self.rank == 0:
            try:
                os.remove(self.path)
            except OSError:
                pass


def cpu_baseline(dr, n_sample: int, m: int, seeds, k: int, tile: int, budget_s: float = 25.0):
    """The oracle's own serial loop (orc_path_process_read: hash, query, decide,
    insert; OpenMP over tiles like the reference) over the first reads of the
    same synthetic set, same filter size; reads/s of its classification phase."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc  # test infrastructure: used here only as the timed CPU baseline
    from goldrush_amd import synth

    orc.build()
    reads = dr.download(0, n_sample)
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"grp_bench_cpu_{os.getpid()}")
    os.makedirs(tmp, exist_ok=True)
    fq = os.path.join(tmp, "sample.fq")
    synth.write_fastq(fq, [(b"r%d" % i, s, b"5" * len(s)) for i, s in enumerate(reads)])
    from goldrush_amd import host

    cores = int(host.load().gr_effective_cpus())  # affinity mask and cgroup quota
    args = ["-k", str(k), "-w16", "-t", str(tile), "-h", str(len(seeds)), "-s", PRESET, "-g", "100000000", "-P10", "-d50", "-m0",
            "-j", str(cores), "-i", fq, "-p", os.path.join(tmp, "o")]
    p = orc.Path(args)
    assert p.ok and p.filter_size() == m
    t0 = time.time()
    done = 0
    for i in range(p.n_reads):
        p.process(i)
        done += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    fill_s, _ = p.timers()
    p.close()
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    return {"value": done / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {done} reads of the same synthetic stream (filter filled with {n_sample} reads, same m={m}); oracle process_read loop "
                      f"(hash + query + decide + insert), OpenMP over tiles, {cores} threads",
            "fill_reads_per_s": n_sample / fill_s if fill_s > 0 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--genome", type=float, default=100e6)
    ap.add_argument("--batch", type=int, default=8192, help="reads per GPU per step")
    ap.add_argument("--h", type=int, default=3)
    ap.add_argument("--max-window", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trace", action="store_true", help="per-step timing / speculation statistics on stderr")
    ap.add_argument("--no-kernel-timing", action="store_true", help="developer: no HIP events around the launches (roofline fields become meaningless)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for single-GPU plumbing tests)")
    ap.add_argument("--verify-ranks", action="store_true", help="developer: N > 1, compare the classifier state of all ranks after the run")
    ap.add_argument("--share-gpu", action="store_true", help="developer: all ranks use GPU 0 (plumbing test on a 1-GPU box, use with --backend gloo)")
    a = ap.parse_args()

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

    G, k, w, tile, h, block = int(a.genome), 22, 16, 1000, a.h, 10
    hl = host.load()
    seeds = host.make_seed_pattern(PRESET, k, w, h)
    m = hl.gr_calc_optimal_size(hl.gr_hash_universe(w, G, h), 1, 0.1)

    # ---- untimed setup: inputs resident in HBM, phase 1 -------------------------
    t0 = time.time()
    dr = native.synth_reads(a.reads, G)
    eng = native.Engine(k, h, tile, m, seeds, device=local_rank)
    rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    t_synth = time.time() - t0
    t0 = time.time()
    if world == 1:
        eng.bv_insert(rb)
        eng.sync()
    else:
        # the fill shards by reads; the merge is a bitwise OR, done as an
        # all-gather of the plain bit vectors (RCCL has no OR reduction) + local ORs
        shard = (a.reads + world - 1) // world
        lo, hi = min(rank * shard, a.reads), min((rank + 1) * shard, a.reads)
        eng.bv_insert(rb, lo, hi - lo)
        eng.sync()
        nbytes = eng.bv_words() * 4
        mine = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        eng.bv_export_device(mine.data_ptr())
        if coll_dev == "cuda":
            allbv = torch.empty(nbytes * world, dtype=torch.uint8, device="cuda")
            dist.all_gather_into_tensor(allbv, mine)
        else:
            tmp = torch.empty(nbytes * world, dtype=torch.uint8)
            dist.all_gather_into_tensor(tmp, mine.cpu())
            allbv = tmp.cuda()
        torch.cuda.synchronize()
        for p in range(world):
            if p != rank:
                eng.bv_merge_device(allbv.data_ptr() + p * nbytes)
        del allbv, mine
    t_fill = time.time() - t0
    fill_stats = eng.kernel_stats()["fill"]
    pop = eng.finalize()

    # ---- phase 2: order-exact classification, windows sharded over the ranks ----
    allgather = None
    shm = None
    if world > 1:
        # The decisions the ranks exchange are tiny (32 B per read, a few KB per call) and
        # already sit in host memory: they go through shared memory (one node) or, failing
        # that, a CPU (gloo) group.  A GPU collective
        # here would need free compute units while the persistent query launch owns the
        # device, and two PCIe copies per call; RCCL is used where bulk data moves (the
        # bit-vector all-gather above).
        if os.path.isdir("/dev/shm") and not os.environ.get("GRP_BENCH_NO_SHM"):
            try:
                shm = ShmAllgather(world, rank, "%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "run")), dist.barrier)
            except Exception as e:
                if rank == 0:
                    sys.stderr.write("bench: no /dev/shm exchange (%s), using a gloo group\n" % e)
                shm = None
        if shm is not None:
            # the exchange costs tens of microseconds here, not hundreds: shorter stripes
            # (less speculative work lost per insert) still hide it behind the launches
            os.environ.setdefault("GRP_STRIPE", str(64 + 8 * world))
        ctrl, ctrl_dev = None, "cpu"
        if shm is None and a.backend == "nccl":
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
            if shm is not None:
                shm(src, dst)
                return 0
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
                          target_bases=int(0.9 * G), max_paths=1, silver_path=False, max_window=a.max_window, world=world, rank=rank,
                          allgather=allgather, record=False)
    per_step = a.batch * world
    n_steps_avail = a.reads // per_step
    if n_steps_avail < 1:
        raise SystemExit("--reads too small for --batch")
    lens = np.ascontiguousarray(dr.lens, dtype=np.uint32)

    def step(i: int):
        # the stream wraps around once exhausted (all of it is then on the path)
        ts = time.perf_counter()
        s0 = cls.state() if a.trace else None
        cls.run_range(rb._h, lens, (i % n_steps_avail) * per_step, per_step)
        if a.trace and rank == 0:
            s1 = cls.state()
            sys.stderr.write("step %d: %.1f ms windows=%d queried=%d inserts=%d\n" % (
                i, (time.perf_counter() - ts) * 1e3, s1["windows"] - s0["windows"], s1["reads_queried"] - s0["reads_queried"], s1["inserts"] - s0["inserts"]))

    if a.no_kernel_timing:
        eng.set_timing(False)
    for i in range(a.warmup):
        step(i)
    eng.sync()
    eng.reset_kernel_stats()
    st0 = cls.state()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
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
        reads_done = a.steps * per_step
        avg_ms = kq["ms"] / max(kq["launches"], 1)
        probes_per_launch = kq["units"] / max(kq["launches"], 1)
        achieved = kq["units"] * 128 / (kq["ms"] * 1e-3) / 1e9 if kq["ms"] > 0 else 0.0
        # HBM bytes per launch: PMC counters cannot be read from inside this process; the
        # per-probe figure comes from the committed rocprofv3 --pmc pass of the same kernel
        # (newest profiles/r*_pmc_summary.json, tools/pmc_round.sh: TCC_EA0_RDREQ x 64 B = FETCH_SIZE x 1024 B)
        traffic = None
        try:
            import glob

            import re

            def _ver(path):  # r01_v10_... sorts after r01_v9_...
                return [int(x) for x in re.findall(r"\d+", os.path.basename(path))]

            newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), key=_ver)[-1]
            pmc = json.load(open(newest))
            if "k_query_all_variants" in pmc:
                traffic = pmc["k_query_all_variants"]["hbm_bytes_per_probe"] * probes_per_launch
            else:
                traffic = pmc["k_query<3, 2, 0>"]["hbm_bytes_per_probe"] * probes_per_launch
        except Exception:
            pass
        out = {
            "metric": "reads/s through GoldRush-Path (hash + miBF query)",
            "value": reads_done / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "C1: 1M synthetic ONT reads, mean 25 kb, G=100e6, k=22 w=16 h=%d tile=1000 o=0.1, golden-path mode, order-exact" % h,
                       "reads": a.reads, "batch_reads_per_gpu": a.batch, "filter_bits": m, "pop": pop,
                       "parallelism": ("one GPU, streaming windows" if world == 1 else "windows striped over %d GPUs (128-read stripes), replicated miBF, 32-B decisions all-gathered per stripe group" % world)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_unit": "bytes per launch (PMC pass of the same kernel, scaled by probes)", "kernel": "k_query", "launches": kq["launches"], "avg_launch_ms": avg_ms,
                         "probes_per_launch": probes_per_launch, "bytes_per_probe": 128},
            "aux": {"fill_reads_per_s": a.reads / t_fill, "fill_Gprobes_per_s": fill_stats["units"] * world / t_fill / 1e9, "fill_s": t_fill,
                    "fill_mode": "single GPU" if world == 1 else "reads sharded over %d GPUs, bit vectors all-gathered and OR-merged" % world, "synth_s": t_synth,
                    "timed": {key: st1[key] - st0[key] for key in ("windows", "reads_queried", "reads_committed", "inserts", "seconds_windows", "seconds_commit")},
                    "query_Gprobes_per_s": kq["units"] / (kq["ms"] * 1e-3) / 1e9 if kq["ms"] > 0 else 0.0,
                    "random_64B_line_ceiling_Gps": 50.0,  # tools/gather_bench.hip on MI355X: 48-55 G random 64-byte lines/s (one line per probe)
                    "query_kernel_s": kq["ms"] * 1e-3, "decide_kernel_s": ks["decide"]["ms"] * 1e-3, "decide_launches": ks["decide"]["launches"],
                    "insert_kernel_s": ks["insert"]["ms"] * 1e-3, "insert_launches": ks["insert"]["launches"],
                    "wall_s": dt},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dr, 1024, m, seeds, k, tile)
        print(json.dumps(out), flush=True)
    if world > 1:
        if shm is not None:
            shm.close(dist.barrier)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
