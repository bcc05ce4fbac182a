#!/usr/bin/env python3
"""GPU box: the drop-in binary end to end.

  tools/cli_end_to_end.py <out.json> [reads] [genome]

Writes a FASTQ of `reads` synthetic ONT-like reads (default 100 000 x 25 kb, G = 100e6: C1's
geometry, ~5 GB of text) and runs goldrush-path on it the way bin/goldrush does — process #1
(--silver_path -M 5 -r 0.9 -m 20000, bin/goldrush:253-260) and process #2 (golden path on the
silver reads, :240-248) — with the reads kept on the device between the passes (default) and with
the second parse (GRP_RESIDENT=off); every run twice, the faster one counts.  Wall time, the program's own phase timers, reads/s
FASTQ-inclusive; bench.py on the same geometry beside it."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLI = os.path.join(ROOT, "goldrush_amd", "bin", "goldrush-path")
SEED = "1011011110110111101101"
REPEATS = int(os.environ.get("CLI_E2E_REPEATS", "2"))


def run_cli(args, env=None):
    t0 = time.perf_counter()
    r = subprocess.run([CLI] + args, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    dt = time.perf_counter() - t0
    ins = [float(x) for x in re.findall(r"^in ([0-9.]+)$", r.stderr, re.M)]
    visited = re.findall(r"Visited (\d+) reads", r.stderr)
    return {"rc": r.returncode, "wall_s": dt, "phase_timers_s": ins, "visited": int(visited[-1]) if visited else None, "stderr_tail": r.stderr[-300:] if r.returncode else ""}


def main():
    out = sys.argv[1]
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    genome = int(float(sys.argv[3])) if len(sys.argv) > 3 else 100_000_000
    from goldrush_amd import synth

    tmp = os.environ.get("TMPDIR", "/tmp")
    fq = os.path.join(tmp, "cli_e2e.fq")
    t0 = time.perf_counter()
    # tools/fqgen.c (OpenMP: the reads of a block made by all cores, written in order) where a C compiler is at hand — the
    # Python generator wrote C1's 51 GB in 227 s, most of a GPU call; the same read model either way
    gen = os.path.join(os.environ.get("FQGEN_DIR", "/tmp"), "fqgen")  # (not TMPDIR: /dev/shm is mounted noexec)
    built = subprocess.run(["gcc", "-O3", "-fopenmp", "-o", gen, os.path.join(ROOT, "tools", "fqgen.c"), "-lm"], capture_output=True).returncode == 0
    if built and subprocess.run([gen, fq, str(n_reads), str(genome), "1"], capture_output=True).returncode == 0:
        generator = "tools/fqgen.c"
    else:
        generator = "goldrush_amd/synth.py"
        g = synth.random_genome(genome, 1)
        with open(fq, "wb") as fh:
            done = 0
            while done < n_reads:  # in slices: the Python objects of 100 k reads would not fit comfortably
                n = min(5000, n_reads - done)
                for rid, seq, qual in synth.make_reads(g, n, mean_len=25000, min_len=20000, seed=2 + done):
                    fh.write(b"@r%d\n" % done + seq + b"\n+\n" + qual + b"\n")
                    done += 1
    size = os.path.getsize(fq)
    res = {"fastq_bytes": size, "reads": n_reads, "genome": genome, "fastq_written_s": time.perf_counter() - t0, "fastq_generator": generator, "runs": {}}
    base = ["-k22", "-w16", "-t1000", "-u5", "-a1", "-o0.1", "-h3", "-j16", "-P10", "-d5", "-x10", "-s" + SEED, "-g%d" % genome, "-b10", "--verbose"]
    pdir = os.path.join(tmp, "cli_e2e_out")
    os.makedirs(pdir, exist_ok=True)
    modes = [m for m in (("resident", {}), ("second_parse", {"GRP_RESIDENT": "off"})) if m[0] in os.environ.get("CLI_E2E_MODES", "resident,second_parse").split(",")]
    for mode, env in modes:
        for name, extra in (("silver_M5", ["-r0.9", "--silver_path", "-M5", "-m20000", "-i", fq, "-p", os.path.join(pdir, "sp_" + mode)]),
                            ("golden_on_raw_reads", ["-m20000", "-i", fq, "-p", os.path.join(pdir, "gp_" + mode)])):
            runs = [run_cli(base + extra, env) for _ in range(REPEATS)]  # the first run on a fresh box also pays the code-object load
            r = min(runs, key=lambda x: x["wall_s"])
            r["wall_s_all"] = [round(x["wall_s"], 3) for x in runs]
            r["reads_per_s_fastq_inclusive"] = (r["visited"] or n_reads) / r["wall_s"]
            r["fastq_GB_per_s"] = size / r["wall_s"] / 1e9
            res["runs"][name + "/" + mode] = r
    # the pipeline's chain (bin/goldrush:239-260): process #1 writes the silver paths, `cat <p1>_*.fq > <p1>_all.fq`,
    # process #2 builds the golden path FROM THAT FILE with -m 0 (round 4)
    sp = sorted(f for f in os.listdir(pdir) if re.fullmatch(r"sp_resident_\d+\.fq", f))
    if sp:
        allfq = os.path.join(pdir, "sp_all.fq")
        t_cat = time.perf_counter()
        with open(allfq, "wb") as dst:
            for f in sp:
                with open(os.path.join(pdir, f), "rb") as src:
                    while True:
                        buf = src.read(1 << 26)
                        if not buf:
                            break
                        dst.write(buf)
        t_cat = time.perf_counter() - t_cat
        n_silver = sum(1 for _ in open(allfq, "rb")) // 4
        runs = [run_cli(base + ["-m0", "-i", allfq, "-p", os.path.join(pdir, "gp_chain")]) for _ in range(REPEATS)]
        r = min(runs, key=lambda x: x["wall_s"])
        r["wall_s_all"] = [round(x["wall_s"], 3) for x in runs]
        r["input_reads"] = n_silver
        r["input_bytes"] = os.path.getsize(allfq)
        r["cat_s"] = t_cat
        r["reads_per_s_fastq_inclusive"] = n_silver / r["wall_s"]
        res["runs"]["golden_on_silver_paths/chain"] = r
        s1 = res["runs"].get("silver_M5/resident")
        if s1:
            res["pipeline_chain"] = {"what": "goldrush-path --silver_path -M 5 on the raw reads, cat of the silver paths, goldrush-path -m 0 on them (bin/goldrush:239-260)",
                                     "raw_reads": n_reads, "silver_reads": n_silver, "seconds": s1["wall_s"] + t_cat + r["wall_s"], "raw_reads_visited_by_process_1": s1["visited"]}
    # the outputs of the two forms are the same files
    same = True
    for f in sorted(os.listdir(pdir)):
        if "_resident" in f and len(modes) == 2:
            a, b = os.path.join(pdir, f), os.path.join(pdir, f.replace("_resident", "_second_parse"))
            same = same and os.path.exists(b) and open(a, "rb").read() == open(b, "rb").read()
    res["outputs_identical"] = same
    # the kernel path alone on the same geometry (GPU-resident synthetic reads, no text)
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C1", "--reads", str(n_reads), "--genome", str(genome), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-pipeline-shaped"], capture_output=True, text=True)
    try:
        line = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
        res["bench_same_geometry"] = {"reads_per_s": line["value"], "wall_s": line["aux"]["wall_s"], "fill_s": line["aux"]["fill_s"], "finalize_s": line["aux"]["finalize_s"]}
    except Exception as e:
        res["bench_same_geometry"] = {"error": str(e), "stderr": b.stderr[-500:]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
    os.remove(fq)


if __name__ == "__main__":
    main()
