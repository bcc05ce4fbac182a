#!/usr/bin/env python3
"""bench.py — GoldRush-Path hot path (spaced-seed ntHash + miBF query) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU with torch.distributed.run.  Rank 0 prints ONE
JSON line.

Workload (BASELINE.json configs[1], "C1"): 1 M synthetic ONT-like reads, mean
25 kb, G = 100e6, k=22 w=16 h=3 tile=1000, o=0.1.  The reads are generated on the
GPU (grp_synth_reads), so all inputs are resident in HBM before the timed region.
A step = one pass of the hot path (hash every tile frame, probe the miBF, per-tile
ID vote) over one batch of `--batch` reads taken in stream order.

metric  : reads/s through hash + miBF query (whole job, all ranks)
roofline: the query kernel, ALGORITHMIC bytes = 128 B per probe (two 64-B HBM
          sectors: bit+rank block, then ID), probes = frames x h; duration from
          HIP events recorded around every launch on the library's own stream.
cpu_baseline: the CPU oracle (restatement of the reference, OpenMP) on a bounded
          sample of the same workload, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PRESET = "1011011110110111101101"  # bin/goldrush:70


def filter_bits(G: int, w: int, h: int, occ: float) -> int:
    """goldrush_path.cpp:1113-1121 (float product) + MIBloomFilter.hpp:94-101."""
    base = min(4 ** w, 2 * G)
    U = int(np.float32(np.float32(base) * np.float32(0.5)) * np.float32(h))
    n = int(-float(U) / np.log(1.0 - occ))
    return n + (64 - n % 64)


def cpu_baseline(dr, n_sample: int, m: int, seeds, k: int, tile: int, h: int):
    """Oracle (CPU restatement, OpenMP over tiles like the reference) timed on a
    bounded sample: fill with the sample, insert every 3rd read as IDs, then time
    hash + query of the sample reads."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc  # test infrastructure: used here only as the timed CPU baseline

    orc.build()
    reads = dr.download(0, n_sample)
    oseeds = orc.Seeds(seeds)
    mf = orc.MiBF(m, oseeds, tile, k)
    for s in reads:
        mf.bv_insert_read(s)
    mf.finalize()
    idn = 0
    for i in range(0, n_sample, 3):
        nt = len(reads[i]) // tile
        idn += 1
        for bs in range(0, nt, 10):
            mf.insert_read_tiles(reads[i], bs, min(bs + 10, nt), idn + bs // 10)
        idn += nt // 10
    lib = orc.load()
    import ctypes as C

    t0 = time.time()
    done = 0
    for s in reads:
        # read_hashing.cpp:29-54 + calc_num_assigned_tiles loop 1, tile by tile
        for t in range(len(s) // tile):
            hv = oseeds.tile_hashes(s, tile, k, t)
            mf.query_tile(hv)
        done += 1
        if time.time() - t0 > 25:
            break
    dt = time.time() - t0
    return {"value": done / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"{done} of the first {n_sample} reads of the same synthetic set, same m; serial oracle (hash+query per tile)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--genome", type=float, default=100e6)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--h", type=int, default=3)
    ap.add_argument("--populate", type=int, default=4000, help="reads whose tiles are inserted as IDs before the timed region (~1x coverage)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from goldrush_amd import native

    G, k, w, tile, h = int(a.genome), 22, 16, 1000, a.h
    seeds = [PRESET[:11] + "0" * i + PRESET[11:] for i in range(h)]
    m = filter_bits(G, w, h, 0.1)

    # ---- untimed setup: inputs resident in HBM ------------------------------
    t0 = time.time()
    dr = native.synth_reads(a.reads, G)
    eng = native.Engine(k, h, tile, m, seeds, device=local_rank)
    rb = eng.wrap_device(dr.d_ptr, dr.word_off, dr.lens)
    t_synth = time.time() - t0
    t0 = time.time()
    eng.bv_insert(rb)
    eng.sync()
    t_fill = time.time() - t0
    fill_stats = eng.kernel_stats()["fill"]
    pop = eng.finalize()
    idn = 0
    for r in range(min(a.populate, a.reads)):
        nt = int(dr.lens[r]) // tile
        idn += 1
        for bs in range(0, nt, 10):
            eng.insert_tiles(rb, r, bs, min(bs + 10, nt), idn + bs // 10)
        idn += nt // 10
    eng.sync()

    # every rank owns a disjoint slice of each step's batch (reads shard; replicated miBF)
    per_rank = a.batch
    n_avail = a.reads - a.populate - per_rank * world
    if n_avail <= 0:
        raise SystemExit("--reads too small for --batch/--populate")

    def step(i: int):
        first = a.populate + ((i * world + rank) * per_rank) % n_avail
        return eng.query_tiles(rb, first, per_rank)

    for i in range(a.warmup):
        step(i)
    eng.reset_kernel_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ks = eng.kernel_stats()["query"]

    if rank == 0:
        reads_done = a.steps * per_rank * world
        avg_ms = ks["ms"] / max(ks["launches"], 1)
        probes_per_launch = ks["units"] / max(ks["launches"], 1)
        achieved = probes_per_launch * 128 / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
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
            "config": {"workload": "C1: 1M synthetic ONT reads, mean 25 kb, G=100e6, k=22 w=16 h=%d tile=1000 o=0.1" % h,
                       "reads": a.reads, "batch_reads_per_gpu": per_rank, "filter_bits": m, "pop": pop,
                       "mode": "query batches against a miBF populated with %d inserted reads" % a.populate,
                       "parallelism": "reads sharded over %d GPU(s), replicated miBF" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": None, "kernel": "k_query", "avg_launch_ms": avg_ms, "probes_per_launch": probes_per_launch,
                         "bytes_per_probe": 128},
            "aux": {"fill_reads_per_s": a.reads / t_fill, "fill_Gprobes_per_s": fill_stats["units"] / t_fill / 1e9,
                    "fill_s": t_fill, "synth_s": t_synth, "query_kernel_reads_per_s": a.steps * per_rank / (ks["ms"] * 1e-3) if ks["ms"] else None},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dr, 48, m, seeds, k, tile, h)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
