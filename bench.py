#!/usr/bin/env python3
"""Headline benchmark: shifts x genome-bp / sec of the per-chromosome strand cross-correlation
(NCC + MaSC) on synthetic hg38-shaped bit-vectors, max_shift = 1000 (BASELINE.json).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (pmx_cc_batch_dev) over ONE synthetic hg38-shaped genome (24 chromosomes,
3.09 Gbp) whose F/R/M bit-vectors are already resident in HBM, the chromosome jobs LPT-sharded over the N ranks
(BASELINE config 4: "sharded 1/2/4/8 GPUs": strong scaling, the default; --scaling weak = N genomes, one per GPU),
followed by the result exchange (all-gather of per-chromosome rows + all-reduce of totals over RCCL), which runs
on its own HIP stream and overlaps the next step's kernels.  Rank 0 prints ONE JSON line.
`python bench.py --gpus N` without a launcher starts the N ranks itself (pymasc_amd/launch.py: fresh child
processes before anything touches the GPU -- what the reference's `-p N` does, handler/calc.py:163-192).
ENCFF000VPI.bam (BASELINE configs 2-3) is not available offline, so the same-shape synthetic workload stands in,
as BASELINE.md section 4 prescribes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: 100 steps of ~0.43 ms are a 43-ms timed region, three of them; regions of 20 steps spread by 4 % on one box)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--shard", choices=["auto", "tiles", "chroms"], default="auto",
                    help="N > 1, strong scaling: equal TILE RANGES of the genome per rank + one all-reduce(sum) of the per-chromosome rows "
                         "(max_shift <= 1023: pmx_cc_batch_ranges_dev), or whole chromosomes by LPT + all-gather; auto: tiles where supported")
    ap.add_argument("--repeat", type=int, default=3,
                    help="repetitions of the timed region of --steps steps; the line reports the median one (min / median / max in `repetitions`)")
    ap.add_argument("--workload", choices=["hg38", "stress"], default="hg38",
                    help="hg38: BASELINE config 4 (default, the metric's configuration); stress: config 5, one "
                         "synthetic 10 Gbp / 200-chromosome genome sharded over the ranks, max_shift 5000")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="hg38 only: strong = ONE genome LPT-sharded over the ranks (BASELINE config 4, default); "
                         "weak = one genome per rank")
    ap.add_argument("--max-shift", type=int, default=None, help="default 1000 (hg38) / 5000 (stress)")
    ap.add_argument("--read-len", type=int, default=None, help="default 36 (hg38) / 100 (stress, BASELINE config 5)")
    ap.add_argument("--density", type=float, default=0.005, help="read start density per strand (BASELINE.md config 4)")
    ap.add_argument("--mode", choices=["both", "ncc"], default="both",
                    help="both = NCC + MaSC with mappability (config 4); ncc = naive CC only (config 2 shape)")
    ap.add_argument("--path", choices=["auto", "dense", "sparse"], default="auto")
    ap.add_argument("--track", choices=["synthetic", "fixture"], default="synthetic",
                    help="mappability track: synthetic = BASELINE.md's geometric runs (mean 2000 on / 500 off); fixture = run "
                         "and gap lengths sampled from the reference's test track hg19_36mer-test.bedGraph (860 run edges "
                         "per 64 Kbit, 23 %% mappable: the only real-track statistics available offline), tiled to the genome")
    ap.add_argument("--chroms", type=int, default=24, help="use only the first K hg38 chromosomes (debug)")
    ap.add_argument("--run-on", type=float, default=2000.0, help="synthetic track: mean mappable run length (BASELINE.md: 2000)")
    ap.add_argument("--run-off", type=float, default=500.0, help="synthetic track: mean gap between runs (BASELINE.md: 500)")
    ap.add_argument("--no-hint", action="store_true",
                    help="do not pass PMX_FLAG_WINDOW_ONLY for dense reads / run edges (what CCHipCalculator derives from the "
                         "counts it holds): the kernels find out by themselves (sweeps)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="one rank only: initialise a 1-rank process group and run the exchange through its collectives "
                         "(RCCL all_gather_into_tensor + all_reduce on the second stream) instead of the local shortcut")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mbp", type=float, default=128.0, help="bp per CPU-baseline slice, in Mbp")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-positions -> rows leg (SURVEY 8d)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the BAM-file leg (SURVEY 8 row f1: device ingest against the host reader)")
    ap.add_argument("--ingest-reads", type=int, default=2_000_000, help="reads of the synthetic BAM file of that leg")
    return ap.parse_args()


def ingest_leg(n_reads, mapq=10):
    """The row in front of the path, outside the timed region and never part of `value`: a synthetic coordinate-sorted
    BAM file of n_reads 36-bp reads over hg38 (BGZF level 1) -> the filtered records of the reference's read loop
    (handler/calc.py:140-153, handler/read.py:62-155), through the HOST reader (zlib on the threads this process may use) and through
    the DEVICE reader (libpymasc_ingest.so: inflate, CRC32, record chain and filter as HIP kernels, records left in HBM); best of
    three each, same records (count and position checksum).  tools/bench_ingest.py is the full-size version (20 M reads)."""
    import tempfile
    from pymasc_amd import bam as B
    from pymasc_amd import bam_device as D
    from tools.bench_ingest import synth_bam
    path = os.path.join(tempfile.gettempdir(), "pymasc_bench_ingest_%d.bam" % os.getpid())
    try:
        synth_bam(path, n_reads)
        cores, _q = usable_cores()

        def host():
            t0 = time.perf_counter()
            n = cs = 0
            with B.BamReader(path, threads=cores) as r:
                for _ref, pos, _rl, rev in r.batches(mapq):
                    n += pos.size
                    cs += int(pos.astype(np.int64).sum()) + int(rev.sum())
            return time.perf_counter() - t0, (n, cs)

        def device():
            t0 = time.perf_counter()
            with D.DeviceBamReader(path) as r:
                n = r.decode(mapq)
                dt = time.perf_counter() - t0      # the filtered records are in HBM (what the device feed consumes)
                cs = 0
                for _ref, pos, _rl, rev in r.batches(mapq):
                    cs += int(pos.astype(np.int64).sum()) + int(rev.sum())
                tm = r.timings()
            return dt, (n, cs), tm
        device()                                   # (the first open page-locks the staging buffers and loads the code object)
        th = [host() for _ in range(3)]
        td = [device() for _ in range(3)]
        assert all(x[1] == th[0][1] for x in th + td), "device ingest and host reader disagree"
        h, d = min(x[0] for x in th), min(x[0] for x in td)
        return {"reads": n_reads, "kept": th[0][1][0], "file_bytes": os.path.getsize(path), "host_reader_s": round(h, 4), "host_threads": cores,
                "device_reader_s": round(d, 4), "speedup": round(h / d, 2),
                "device_phases_s": {k: round(v, 4) for k, v in min(td, key=lambda x: x[0])[2].items()},
                "what": "BAM file -> filtered (ref, pos, read length, strand) records: zlib host reader vs libpymasc_ingest.so "
                        "(records left in HBM); same records; not part of `value`"}
    except Exception as e:          # the contract line must not depend on this leg
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        if os.path.exists(path):
            os.unlink(path)


def ingest_leg_isolated(n_reads):
    """ingest_leg in a child process (started fresh: `python bench.py --ingest-child N`, not an exec of this one): whatever happens to
    that leg -- an exception, a device fault that ends its process -- the contract line of this process is still printed."""
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--ingest-child", str(int(n_reads))], capture_output=True, text=True,
                           timeout=240)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": "the ingest leg's process ended with code %d: %s" % (p.returncode, (p.stderr or "").strip().splitlines()[-1:] or "")}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def cpu_baseline(ctx, vecs, S, L, with_m, sample_bp, threads, budget_s=14.0):
    """Times the oracle (C restatement of the reference's per-shift full-vector passes) on host cores:
    one chromosome slice per thread, like `pymasc -p <threads>` (threads = the cores this process
    may run on).  The sample is bounded by wall-clock: a short calibration slice on one thread sets the slice length so
    that the threaded run takes about `budget_s` seconds.  kind = "port"."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import model as oracle
    oracle.lib()

    def make_slice(i, bp):
        v = vecs[i % len(vecs)]
        nb = int(min(v.length, bp)) + L + S + 100
        nw = (nb + 63) // 64
        F = ctx.bits_download(v.F.data_ptr(), nb)[:nw].copy()
        R = ctx.bits_download(v.R.data_ptr(), nb)[:nw].copy()
        M = ctx.bits_download(v.M.data_ptr(), nb)[:nw].copy() if with_m else None
        top = nb & 63
        if top:
            mask = np.uint64((1 << top) - 1)
            F[-1] &= mask
            R[-1] &= mask
            if M is not None:
                M[-1] &= mask
        return (F, R, M, nb, nb - (L + S + 100))

    # calibration: one small slice alone on one core (also the 1-thread figure SURVEY 8d asks for)
    s0 = make_slice(0, min(sample_bp, 8e6))
    t1 = time.perf_counter()
    oracle.calc_correlation(s0[0], s0[1], s0[2], s0[3], S, L)
    dt1 = time.perf_counter() - t1
    one_thread = (S + 1) * s0[4] / dt1
    # all threads run one slice each at the same time: the wall time is one slice's (memory bandwidth permitting,
    # assume half the single-thread rate under load)
    slice_bp = min(sample_bp, max(1e6, 0.5 * one_thread / (S + 1) * budget_s))
    slices = [make_slice(i, slice_bp) for i in range(threads)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda s: oracle.calc_correlation(s[0], s[1], s[2], s[3], S, L), slices))
    dt = time.perf_counter() - t0
    work = (S + 1) * sum(s[4] for s in slices)
    return {"value": work / dt, "unit": "shifts*bp/s", "cores": threads, "kind": "port", "sampled": True,
            "cpu_model": cpu_model(), "nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
            "cgroup_cpu_quota": usable_cores()[1],
            "compiler": "gcc " + " ".join(oracle.ORACLE_CFLAGS) + " oracle/cc_oracle.c",
            "one_thread_value": one_thread,
            "sample": f"SAMPLE, not the whole genome: {threads} slices x {slices[0][4] / 1e6:.1f} Mbp of the same synthetic chromosomes, "
                      f"{'NCC+MSCC' if with_m else 'NCC'}, max_shift={S}, one slice per thread on the {threads} cores this "
                      f"process may use (nproc {os.cpu_count()}), {dt:.1f}s wall",
            "seconds": dt}


def usable_cores():
    """Cores this process may use: its affinity mask, cut by the cgroup's CPU quota where one is set (a GPU box hands a
    job a share of the host: threads beyond the quota only time-slice)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


PROFILE_ROUND = "r4"


def committed_profile(name, workload_tag):
    """A counter summary under profiles/ (tools/tools_r4_profile.sh), or (None, reason).  The summaries name the build
    (pmx_build_id: hash of pymasc_amd/csrc + the header) and the bench workload they were measured on; they are quoted only
    for that build and that workload, so a kernel edit cannot leave stale HBM bytes or pipe utilisation in the line."""
    from pymasc_amd import ffi
    import glob
    why = f"no profiles/{PROFILE_ROUND}_{name}*.json"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{name}*.json"))):
        rel = os.path.relpath(path, ROOT)
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        if prof.get("workload") != workload_tag:
            if why.startswith("no "):
                why = f"no profiles/{PROFILE_ROUND}_{name}*.json for workload '{workload_tag}'"
            continue
        if prof.get("build_id") != ffi.build_id():
            why = f"{rel} was measured on build {prof.get('build_id')}, the loaded library is build {ffi.build_id()}"
            continue
        return prof, rel
    return None, why


def pipe_utilisation(kernel, workload_tag):
    """Measured VALU-issue / LDS-pipe utilisation of the dominant kernel (rocprofv3 --pmc passes of this same command)."""
    prof, why = committed_profile("pmc_summary", workload_tag)
    if prof is None:
        return {"valu_issue": None, "lds_pipe": None, "source": None, "reason": why}
    k = prof["kernels"].get(kernel.split("+")[0])
    if not k:
        return {"valu_issue": None, "lds_pipe": None, "source": None, "reason": f"{why} holds no entry for {kernel}"}
    return {"valu_issue": k["valu_issue_util"], "lds_pipe": k["lds_pipe_util"],
            "lds_bank_conflict_frac": k.get("lds_bank_conflict_frac"),
            "wave_cycles_issuing": k.get("wave_cycles_issuing"), "wave_cycles_waiting": k.get("wave_cycles_waiting"),
            "source": why + " (" + prof.get("how", "rocprofv3 --pmc") + ")", "build_id": prof.get("build_id")}


def issue_rate(kernel, workload_tag):
    """Issued instructions per SIMD-cycle of the dominant kernel: every instruction of a wave's stream (vector, scalar, LDS,
    branch, memory) counted by the SQ counters, x waves, / (kernel duration x 2.4 GHz x 1024 SIMDs).  DESIGN.md section 8: the
    event kernel's time follows this count whatever the instruction type; a SIMD issues ~0.4 wave64 instructions per cycle
    at best (tools/valu_rate.hip: 2.5 cycles per full-rate vector instruction)."""
    prof, why = committed_profile("pmc_summary", workload_tag)
    if prof is None:
        return {"insts_per_simd_cycle": None, "reason": why}
    k = prof["kernels"].get(kernel.split("+")[0])
    if not k or not k.get("waves"):
        return {"insts_per_simd_cycle": None, "reason": f"{why} holds no entry for {kernel}"}
    per_wave = k["insts_per_wave"]
    kinds = ("valu", "salu", "lds", "branch", "vmem_rd", "vmem_wr")
    total = sum(per_wave.get(x, 0.0) for x in kinds) * k["waves"]
    cycles = k["avg_duration_us"] * 1e-6 * 2.4e9 * 1024
    return {"insts_per_simd_cycle": total / cycles, "insts_per_wave": {x: per_wave.get(x) for x in kinds},
            "peak_insts_per_simd_cycle": 0.4, "frac_of_issue_peak": total / cycles / 0.4,
            "source": why + " (SQ_INSTS_* per dispatch; duration from the --stats pass of the same command)", "build_id": prof.get("build_id")}


def main():
    args = parse()
    from pymasc_amd import launch
    if launch.needs_spawn(args.gpus):
        # `python bench.py --gpus N` without a launcher: start the N ranks here, BEFORE anything touches the GPU
        # (fresh child processes, the reference's `-p N`: handler/calc.py:163-192); rank 0 prints the JSON line
        # (a rank that hangs in a collective must not hang the launcher for ever: a generous limit, exit code 124)
        sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                    timeout=float(os.environ.get("BENCH_LAUNCH_TIMEOUT", "1500"))))

    import torch
    import torch.distributed as dist
    from pymasc_amd import ffi, sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # BENCH_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices, the
    # exchange is staged through host memory); the driver's multi-GPU runs use the default: nccl = RCCL over xGMI.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    elif args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": device} if backend == "nccl" else {}))

    stress = args.workload == "stress"
    strong = stress or args.scaling == "strong"
    if args.max_shift is None:
        args.max_shift = 5000 if stress else 1000
    if args.read_len is None:
        args.read_len = 100 if stress else 36
    S, L = args.max_shift, args.read_len
    with_m = args.mode == "both"
    flags = {"auto": 0, "dense": ffi.PMX_FLAG_FORCE_DENSE, "sparse": ffi.PMX_FLAG_FORCE_SPARSE}[args.path]
    # Two streams: the vector builders and the hot-path kernels run on `tstream` (the context's stream, where the
    # HIP-event kernel timing is taken); the result exchange (gather index ops / RCCL collectives) runs on `xstream`
    # and overlaps the next step's kernels.  Steps queue back to back without host synchronisation between them.
    tstream = torch.cuda.Stream(device)
    xstream = torch.cuda.Stream(device)
    torch.cuda.set_stream(tstream)
    ctx = ffi.Context(dev_index, stream=tstream.cuda_stream)

    chroms = synth.stress_genome() if stress else synth.HG38[:args.chroms]
    # strong: ONE genome LPT-sharded over all ranks (BASELINE configs 4 and 5); weak: `world` genomes, one per rank.
    # LPT over ranks (identical on every rank)
    nsamples = 1 if strong else world
    jobs = [(s, i) for s in range(nsamples) for i in range(len(chroms))]
    costs = [chroms[i][1] for (_s, i) in jobs]
    assignment = sharding.lpt_assign(costs, world)
    # Tile ranges (round 4): the genome's 64-Kbit tiles in `world` equal stretches -- a rank holds whole chromosomes and a share of
    # at most two --, partial result blocks, ONE all-reduce(sum) (BASELINE.json's north star).  Needs the event kernel of
    # max_shift <= 1023 (pmx_cc_batch_ranges_dev); otherwise whole chromosomes by LPT and an all-gather, as in rounds 1-3.
    job_nbits = [chroms[i][1] + L + S + 100 for (_s, i) in jobs]            # (synth.make_chromosome)
    tiles_ok = strong and S <= 1023 and 3 <= S and L <= 1024 and args.path == "auto"
    shard_tiles = world > 1 and (args.shard == "tiles" or (args.shard == "auto" and tiles_ok))
    if shard_tiles and not tiles_ok:
        raise SystemExit("--shard tiles needs strong scaling, 3 <= max_shift <= 1023, read_len <= 1024 and --path auto")
    ranges = sharding.tile_range_assign(job_nbits, world, ffi.RANGE_TILE_BITS) if shard_tiles else None
    mine = [j for j, _f, _c in ranges[rank]] if shard_tiles else assignment[rank]
    max_slots = max(len(a) for a in (ranges if shard_tiles else assignment))

    e2e = not args.no_end_to_end
    t_gen = time.perf_counter()
    vecs = []
    for j in mine:
        s, i = jobs[j]
        name, length = chroms[i]
        vecs.append(synth.make_chromosome(ctx, device, f"{name}.s{s}", length, S, L, 0xC0FFEE + i + 1000 * s,
                                          density=args.density, with_m=with_m, keep_host=e2e, track=args.track,
                                          mean_on=args.run_on, mean_off=args.run_off))
    t_gen = time.perf_counter() - t_gen
    total_bp = sum(costs)

    stride = S + 1
    d_rows = [torch.zeros((max_slots, ffi.PMX_NROWS, stride), dtype=torch.int64, device=device) for _ in range(2)]

    pF = [v.F.data_ptr() for v in vecs]
    pR = [v.R.data_ptr() for v in vecs]
    pM = [v.M.data_ptr() for v in vecs] if with_m else None
    pN = [v.nbits for v in vecs]
    pO = [[buf[slot].data_ptr() for slot in range(len(vecs))] for buf in d_rows]
    ev_done = [torch.cuda.Event() for _ in range(2)]     # kernels of the step wrote d_rows[b]
    ev_free = [torch.cuda.Event() for _ in range(2)]     # the exchange of the step has read d_rows[b]
    nstep = [0]

    # the hint the calculator gives from the read counts it holds (pymasc_amd/calculator.py: window_only_hint)
    from pymasc_amd.calculator import window_only_hint, deep_lists_hint
    dense = any(window_only_hint(v.n_forward, v.n_reverse, v.n_runs if with_m else 0, v.length, S) for v in vecs)
    deep = not dense and any(deep_lists_hint(v.n_forward, v.n_reverse, v.n_runs if with_m else 0, v.length, S) for v in vecs)
    hinted = (dense or deep) and args.path == "auto" and not args.no_hint
    # (--no-hint: flags = 0, and pmx_cc_batch_dev takes the hint itself from a sample of the vectors: one small launch + one
    # synchronisation per call; otherwise one of the three hints, as the calculator always gives one)
    step_flags = flags | ((ffi.PMX_FLAG_WINDOW_ONLY if dense else ffi.PMX_FLAG_DEEP_LISTS) if hinted else
                          (ffi.PMX_FLAG_EVENTS_HINT if args.path == "auto" and not args.no_hint else 0))

    xtimes = []          # (start, end) events of every step's result exchange on xstream

    if shard_tiles:
        t_first = [f for _j, f, _c in ranges[rank]]
        t_count = [c for _j, _f, c in ranges[rank]]
        d_full = [torch.zeros((len(jobs), ffi.PMX_NROWS, stride), dtype=torch.int64, device=device) for _ in range(2)]
        my_jobs = torch.tensor(mine, dtype=torch.int64, device=device)

    def step():
        # all of this rank's chromosomes in ONE pass of the kernels (pmx_cc_batch_dev) on tstream, then the exchange
        # on xstream (double-buffered result blocks)
        b = nstep[0] & 1
        if nstep[0] >= 2:
            tstream.wait_event(ev_free[b])
        nstep[0] += 1
        if vecs and shard_tiles:
            ctx.cc_batch_ranges_dev(pF, pR, pM, pN, t_first, t_count, S, L, step_flags & ~ffi.PMX_FLAG_EVENTS_HINT, pO[b])
        elif vecs:
            ctx.cc_batch_dev(pF, pR, pM, pN, S, L, step_flags, pO[b])
        ev_done[b].record(tstream)
        with torch.cuda.stream(xstream):
            xstream.wait_event(ev_done[b])
            x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            x0.record(xstream)
            if shard_tiles:
                # my shares into a zeroed [jobs, rows, shifts] block, ONE all-reduce(sum): every rank holds every chromosome's rows
                d_full[b].zero_()
                if vecs:
                    d_full[b].index_copy_(0, my_jobs, d_rows[b][:len(mine)])
                if backend != "nccl":
                    rows_ = sharding.exchange_partial_rows(d_full[b].cpu())
                else:
                    rows_ = sharding.exchange_partial_rows(d_full[b])
                out = (rows_, rows_.sum(dim=0))
            elif world > 1 and backend != "nccl":
                out = sharding.exchange_results(d_rows[b].cpu(), assignment, len(jobs))
            else:
                out = sharding.exchange_results(d_rows[b], assignment, len(jobs), force_collectives=args.force_collectives)
            x1.record(xstream)
            xtimes.append((x0, x1))
            ev_free[b].record(xstream)
        return out

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    # Live kernel timing costs a HIP-event pair (~6 us of stream time) per bracketed launch.  One extra untimed step at
    # level 2 shows whether the window-kernel launches behind the event kernel do any work on this workload (dense
    # tiles); if they do not (they return at once), the timed region brackets only the kernels that work (level 1).
    ctx.set_profiling(2)
    ctx.reset_kernel_times()
    step()
    fence()
    probe = {k: ctx.kernel_time(k)[0] for k in range(ffi.PMX_KERNEL_COUNT)}
    fallback_ms = probe[ffi.PMX_KERNEL_CC_SPARSE] + probe[ffi.PMX_KERNEL_AUTOCORR]
    # (two empty launches take 10-20 us together, whatever the shard size; real work on dense tiles takes milliseconds)
    prof_level = 2 if (probe[ffi.PMX_KERNEL_CC_EVENTS] > 0 and fallback_ms > 0.04) else 1
    ctx.set_profiling(prof_level)
    # The timed region: EXACTLY args.steps steps between two fences (barrier + synchronize on both sides), max over ranks.
    # It is run `args.repeat` times (default 3) and the line reports the MEDIAN repetition -- elapsed, ms_per_step, value and
    # the live kernel times all belong to that one repetition of args.steps steps; min / median / max go into `repetitions`
    # (a 10-ms region on one box spreads by a few per cent, more than some of the A/B differences DESIGN.md books).
    reps = []
    for _ in range(max(args.repeat, 1)):
        ctx.reset_kernel_times()
        del xtimes[:]
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rows, totals = step()
        fence()
        reps.append((time.perf_counter() - t0, {k: ctx.kernel_time(k) for k in range(ffi.PMX_KERNEL_COUNT)},
                     sum(a.elapsed_time(b) for a, b in xtimes) / max(len(xtimes), 1)))
    ctx.set_profiling(False)
    if world > 1:
        t = torch.tensor([r[0] for r in reps], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reps = [(float(x), r[1], r[2]) for x, r in zip(t.tolist(), reps)]
    order = sorted(range(len(reps)), key=lambda i: reps[i][0])
    elapsed, ktimes, exchange_ms = reps[order[len(order) // 2]]
    rep_ms = sorted(1e3 * r[0] / args.steps for r in reps)

    # consistency of the exchange: all-reduced totals == sum of gathered rows (integers, exact)
    assert torch.equal(rows.sum(dim=0), totals), "result exchange mismatch"
    if shard_tiles:
        # the shares must add up to whole chromosomes: every block carries ONE path marker (the share that holds tile 0 writes
        # it), and this rank's whole chromosomes must equal a plain pmx_cc_batch_dev run of them
        assert bool((rows[:, ffi.PMX_ROW_SCALARS, 3] == ffi.PMX_PATH_SPARSE).all()), "tile-range shares do not add up (path marker)"
        whole = [k for k, (j, f, c) in enumerate(ranges[rank]) if c == (job_nbits[j] + ffi.RANGE_TILE_BITS - 1) // ffi.RANGE_TILE_BITS]
        if whole:
            k = whole[0]
            chk = torch.zeros((ffi.PMX_NROWS, stride), dtype=torch.int64, device=device)
            ctx.cc_batch_dev([pF[k]], [pR[k]], [pM[k]] if with_m else None, [pN[k]], S, L, step_flags, [chk.data_ptr()])
            ctx.sync()
            assert torch.equal(chk.to(rows.device), rows[mine[k]]), "tile-range rows differ from the whole-chromosome run"

    # dominant kernel + roofline from the live HIP-event timings on this rank's stream
    dom = max(ktimes, key=lambda k: ktimes[k][0])
    dom_ms, dom_n = ktimes[dom]
    vec_bytes = sum((v.nbits + 7) // 8 for v in vecs)
    out_bytes = 8 * stride
    per_pass = {
        ffi.PMX_KERNEL_CC_DENSE: (3 if with_m else 2) * vec_bytes + (4 if with_m else 1) * out_bytes * len(vecs),
        ffi.PMX_KERNEL_CC_SPARSE: (3 if with_m else 2) * vec_bytes + (4 if with_m else 1) * out_bytes * len(vecs),
        ffi.PMX_KERNEL_AUTOCORR: vec_bytes + out_bytes * len(vecs),
        ffi.PMX_KERNEL_CC_EVENTS: (3 if with_m else 2) * vec_bytes + (4 if with_m else 1) * out_bytes * len(vecs),
    }[dom]
    alg_bytes_per_launch = per_pass * args.steps / max(dom_n, 1)
    dense_lane_ops = (S + 1) * (8 * vec_bytes / 32) * (9 if with_m else 3) * args.steps / max(dom_n, 1)
    avg_ms = dom_ms / max(dom_n, 1)
    achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    kernel_ms_per_step = {ctx.kernel_name(k): round(ktimes[k][0] / args.steps, 4) for k in ktimes if ktimes[k][1]}

    # HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate
    # rocprofv3 passes of this same command: tools/tools_r4_profile.sh -> profiles/r4_traffic*.json).  Quoted only when
    # the summary was measured on THIS build of the library and on this workload.
    workload_tag = (f"{args.workload}/{args.mode}/S{S}/L{L}/rho{args.density}/chroms{len(chroms)}/path{args.path}/"
                    f"track{args.track}" + ("" if (args.run_on, args.run_off) == (2000.0, 500.0) else f"-{args.run_on:g}-{args.run_off:g}")
                    + ("/nohint" if args.no_hint else "") + f"/gpus{world}")
    traffic, traffic_src = None, None
    prof, why = committed_profile("traffic", workload_tag)
    if prof is not None:
        key = {ffi.PMX_KERNEL_CC_SPARSE: "k_cc_sparse", ffi.PMX_KERNEL_AUTOCORR: "k_autocorr_pairs",
               ffi.PMX_KERNEL_CC_EVENTS: "k_cc_events"}.get(dom)
        if key in prof.get("kernels", {}):
            traffic = prof["kernels"][key]["hbm_bytes"]
            traffic_src = why + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; build " + prof["build_id"] + ")"
        else:
            traffic_src = f"{why} holds no entry for {key}"
    else:
        traffic_src = why

    work_per_step = (S + 1) * total_bp
    value = work_per_step * args.steps / elapsed

    # what every rank did, in the line of rank 0 (N > 1): one driver run explains itself -- the share of the genome each rank
    # holds (LPT over whole chromosomes), its kernels' time per step, the result exchange's time on its second stream
    loads = ([sum(c for _j, _f, c in r) * ffi.RANGE_TILE_BITS for r in ranges] if shard_tiles else
             [sum(costs[j] for j in a) for a in assignment])
    mine_info = {"rank": rank, "jobs": len(mine), "bp": int(loads[rank]),
                 "kernel_ms_per_step": {ctx.kernel_name(k): round(ktimes[k][0] / args.steps, 4) for k in ktimes if ktimes[k][1]},
                 "exchange_ms_per_step": round(exchange_ms, 4)}
    per_rank = [mine_info]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_info)
    multi_gpu = {"ranks": world, "lpt_imbalance": (max(loads) / (sum(loads) / world) - 1.0) if sum(loads) else 0.0,
                 "sharding": ("tile ranges: the genome's 64-Kbit tiles in equal stretches (pmx_cc_batch_ranges_dev)" if shard_tiles
                              else "whole chromosomes, longest first (LPT)"),
                 "exchange": ("ONE all_reduce(sum) of the per-chromosome rows" if shard_tiles else
                              "all_gather_into_tensor(rows) + all_reduce(totals)") + " on a second stream, overlapping the next step's kernels",
                 "per_rank": per_rank,
                 "measured_on_hardware": bool(world > 1 and backend == "nccl")}

    # ---- end to end: reads + intervals in HOST memory -> vectors -> the same kernels -> rows back in host
    # memory.  Never `value`; reported beside it.  Two legs:
    #   end_to_end            bit positions / intervals (uint32, page-locked host arrays) -> pmx_bits_build_batch (ONE
    #                         stream-ordered call per genome, copies one chromosome ahead of the builder kernels, range check
    #                         deferred) -> pmx_cc_batch_dev -> exchange -> rows
    #   end_to_end_calculator the drop-in boundary itself: CCHipCalculator.feed_reads per chromosome (reads in file order:
    #                         int32 position with the strand in its top bit, read length as one int or uint16; page-locked) -> finishup_calculation ->
    #                         get_whole_result (the reference's result objects), rows compared with the resident-vector run
    end_to_end = None
    calc_leg = None
    if e2e:
        def pinned(a, dt):
            out = ctx.host_array(a.size, dt)
            out[:] = a
            return out
        # one page-locked block per chromosome, the four arrays back to back (Context.host_packed): one copy per chromosome
        host = []
        for v in vecs:
            src = [v.h_fpos, v.h_rpos] + ([v.h_first, v.h_last] if with_m else [])
            dst = ctx.host_packed([a.size for a in src], np.uint32)
            for d_, a in zip(dst, src):
                d_[:] = a
            host.append(tuple(dst) + ((None, None) if not with_m else ()))
        build_jobs = [(v.F.data_ptr(), v.R.data_ptr(), v.M.data_ptr() if with_m else None, v.nbits, h[0], h[1], h[2], h[3])
                      for v, h in zip(vecs, host)]

        def e2e_step():
            keep = ctx.bits_build_batch(build_jobs, np.uint32)
            rws, _tot = step()
            xstream.synchronize()
            out = rws.cpu()
            ctx.bits_build_status()              # the deferred range check of the whole genome (one read-back)
            del keep
            return out
        e2e_step()
        fence()
        n_e2e = 5
        t1 = time.perf_counter()
        for _ in range(n_e2e):
            host_rows = e2e_step()
        fence()
        dt = (time.perf_counter() - t1) / n_e2e
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if not torch.equal(host_rows, rows.cpu()):
            bad = (host_rows != rows.cpu()).nonzero()
            raise AssertionError("end-to-end rows differ from the resident-vector rows: %d cells, first %s (e2e %s, resident %s); shape %s"
                                 % (bad.shape[0], bad[0].tolist(), host_rows[tuple(bad[0].tolist())].item(),
                                    rows.cpu()[tuple(bad[0].tolist())].item(), tuple(host_rows.shape)))
        h2d = sum(h[0].nbytes + h[1].nbytes + ((h[2].nbytes + h[3].nbytes) if with_m else 0) for h in host)
        end_to_end = {"value": work_per_step / dt, "unit": "shifts*bp/s", "ms_per_step": dt * 1e3, "steps": n_e2e,
                      "h2d_bytes_this_rank": h2d, "d2h_bytes": int(host_rows.numel() * 8),
                      "what": "uint32 bit positions + intervals in page-locked host memory (one block per chromosome) -> pmx_bits_build_batch (one "
                              "stream-ordered call per genome: H2D copies one chromosome ahead of the builder kernels, deferred "
                              "range check) -> k_cc_events (+ window kernels for dense tiles) -> exchange -> rows in host memory"}

        if world == 1:
            # ---- the calculator leg.  Reads in file order per chromosome: forward reads at their bit, reverse reads at
            # bit - len + 1 with len = read_len (shorter at the chromosome's first bases), merged by position (stable).
            from pymasc_amd.calculator import CCHipCalculator

            class TrackFeeder:       # what CCHipCalculator asks of a BigWig reader (pymasc_amd/bigwig.py: fetch_arrays)
                def __init__(self, tracks):
                    self.tracks = tracks

                def fetch_arrays(self, _threshold, chrom):
                    return self.tracks[chrom]

            reads, tracks = {}, {}
            for v, h in zip(vecs, host):
                fb, rb = np.unique(v.h_fpos), np.unique(v.h_rpos)
                rl = np.minimum(L, rb).astype(np.int64)
                pos = np.concatenate([fb, rb - rl + 1])
                ln = np.concatenate([np.full(fb.size, L, dtype=np.int64), rl])
                rv = np.concatenate([np.zeros(fb.size, dtype=np.uint8), np.ones(rb.size, dtype=np.uint8)])
                order = np.argsort(pos, kind="stable")
                # a run of reads of one length passes that length as a scalar (what a reader of single-end ChIP-seq data sees)
                uniform = bool((ln == L).all())
                # ... and the strand travels in the top bit of the position word (ffi.pack_strand): 4 bytes per read
                # BENCH_FEED_FORMAT=pos32 (A/B: round 3's form): ffi.pack_strand, 4 bytes per read
                if os.environ.get("BENCH_FEED_FORMAT") == "pos32":
                    reads[v.name] = (pinned(ffi.pack_strand(pos[order].astype(np.int32), rv[order]), np.int32),
                                     L if uniform else pinned(ln[order], np.uint16), None)
                else:   # ... or, since the reads are sorted, as distances: TWO bytes per read (ffi.pack_delta16), one page-locked block
                    d16 = ffi.pack_delta16(pos[order], rv[order], ctx, None if uniform else ln[order])
                    reads[v.name] = (d16, L if uniform else d16.readlen, None)
                if with_m:      # BigWig (begin, end): set(begin + 1, end); both arrays in one page-locked block
                    b_, e_ = ctx.host_packed([v.h_first.size, v.h_last.size], np.uint32)
                    b_[:] = v.h_first - 1
                    e_[:] = v.h_last
                    tracks[v.name] = (b_, e_, None)
            names = [v.name for v in vecs]
            lens = [v.length for v in vecs]

            stamps = []
            trace_acc = {}

            def _timed(obj, nm):
                f = getattr(obj, nm)
                if getattr(f, "_timed", False):
                    return

                def g(*a, **kw):
                    t = time.perf_counter()
                    try:
                        return f(*a, **kw)
                    finally:
                        e = trace_acc.setdefault(nm, [0, 0.0])
                        e[0] += 1
                        e[1] += time.perf_counter() - t
                g._timed = True
                setattr(obj, nm, g)

            def calc_step():
                t0 = time.perf_counter()
                calc = CCHipCalculator(S, L, names, lens, bwfeeder=TrackFeeder(tracks) if with_m else None, context=ctx)
                if os.environ.get("BENCH_EARLY_BATCH"):      # (A/B: launch the kernels of every N queued chromosomes during the feed)
                    calc.early_batch = int(os.environ["BENCH_EARLY_BATCH"])
                if os.environ.get("BENCH_CALC_TRACE"):       # (where a genome's host time goes, per method: stderr)
                    for nm in ("_run_cc", "_materialize", "_calc_correlation", "_load_mappability", "_to_device", "_fill_result"):
                        _timed(calc, nm)
                    _timed(ctx, "bits_download")
                    _timed(ctx, "cc_batch_dev")
                    _timed(ctx, "feed_reads_delta16")
                    _timed(ctx, "bits_set_regions_async")
                if os.environ.get("BENCH_TRACK"):            # (A/B: "plain" = track cleared and OR-ed into on the context's stream,
                    mode = os.environ["BENCH_TRACK"]         #  "side" = that on the side stream, "build" = built whole on the context's stream)
                    calc.track_side_stream = mode in ("side", "both")
                    calc.track_builder = mode in ("build", "both")
                tc = time.perf_counter()
                for v in vecs:
                    calc.feed_reads(v.name, *reads[v.name])
                t1 = time.perf_counter()
                if os.environ.get("BENCH_CALC_PROFILE") == "2" and len(stamps) == 2:
                    import cProfile
                    import pstats
                    pr = cProfile.Profile()
                    pr.enable()
                    calc.finishup_calculation()
                    pr.disable()
                    pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
                else:
                    calc.finishup_calculation()
                tf = time.perf_counter()
                whole = calc.get_whole_result()
                t2 = time.perf_counter()
                calc.close()
                stamps.append((tc - t0, t1 - tc, tf - t1, t2 - tf, time.perf_counter() - t2))
                return whole
            # (the interpreter's generational garbage collector walks every live object of this process -- torch included --
            # whenever its allocation counters trip: ~45 ms once every few calls here, nothing to do with the path measured;
            # it is paused for the timed calls, as for any micro-benchmark of Python-level code.  Collected BEFORE the untimed
            # genomes: the card idles through a collection, and the first genomes after an idle stretch run 10-20 % slower)
            import gc
            gc.collect()
            gc.disable()
            for _ in range(4):      # (untimed genomes: staging slots and vector pool reach their sizes, the card is back at its clocks)
                calc_step()
            fence()
            if os.environ.get("BENCH_CALC_PROFILE"):      # where the host time of the calculator leg goes (stderr)
                import cProfile
                import pstats
                pr = cProfile.Profile()
                pr.enable()
                calc_step()
                fence()
                pr.disable()
                pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(35)
            trace_acc.clear()
            del stamps[:]
            t1 = time.perf_counter()
            for _ in range(n_e2e):
                whole = calc_step()
            fence()
            dtc = (time.perf_counter() - t1) / n_e2e
            gc.enable()
            if trace_acc:
                ncall = len(stamps)
                print("[calc trace] ms per genome (inclusive):", {k: (v[0] // ncall, round(1e3 * v[1] / ncall, 3)) for k, v in trace_acc.items()},
                      file=sys.stderr)
            print("[calc leg] construct / feed (host, paced by the copies) / finishup / get_whole_result / close (ms), every call:",
                  [[round(x * 1e3, 2) for x in st] for st in stamps], file=sys.stderr)
            # rows equal to the resident-vector run (job order = this rank's slot order at one rank)
            hr = rows.cpu().numpy()
            for slot, j in enumerate(mine):
                name = vecs[slot].name
                assert np.array_equal(whole.chroms[name].ccbins, hr[j, ffi.PMX_ROW_NCC_CCBINS]), "calculator ncc differs"
                assert whole.chroms[name].forward_sum == int(hr[j, ffi.PMX_ROW_SCALARS, 0])
                if with_m:
                    mc = whole.mappable_chroms[name]
                    assert np.array_equal(mc.ccbins, hr[j, ffi.PMX_ROW_MSCC_CCBINS]), "calculator mscc.ccbins differ"
                    assert np.array_equal(mc.forward_sum, hr[j, ffi.PMX_ROW_MSCC_FSUM]) and np.array_equal(mc.reverse_sum, hr[j, ffi.PMX_ROW_MSCC_RSUM])
            nreads = sum(r[0].size for r in reads.values())
            d16_mode = os.environ.get("BENCH_FEED_FORMAT") != "pos32"
            read_bytes = lambda r: ((r[0].words.nbytes + r[0].seg_start.nbytes + r[0].seg_base.nbytes) if d16_mode else r[0].nbytes)
            calc_leg = {"value": work_per_step / dtc, "unit": "shifts*bp/s", "ms_per_step": dtc * 1e3, "steps": n_e2e,
                        "reads": int(nreads), "reads_per_s": nreads / dtc,
                        "read_format": "delta16 (2 bytes per read)" if d16_mode else "pos32 (4 bytes per read)",
                        "h2d_bytes": int(sum(read_bytes(r) + getattr(r[1], "nbytes", 0) + getattr(r[2], "nbytes", 0) for r in reads.values())
                                         + sum(t[0].nbytes + t[1].nbytes for t in tracks.values())),
                        "what": "CCHipCalculator (the class handler/factory.py constructs): feed_reads(chrom, reads in file order as 16-bit "
                                "words (strand + distance to the read before, ffi.pack_delta16; BENCH_FEED_FORMAT=pos32: int32 positions with "
                                "the strand in the top bit), read length (one int per chromosome where all reads have it, else uint16); "
                                "one page-locked block per chromosome) -> pmx_feed_reads_delta16 / pmx_feed_reads "
                                "(duplicate rules + read-length sums + bit set on the device) -> finishup_calculation (one batched "
                                "pmx_cc_batch_dev, one synchronisation, one copy back) -> get_whole_result (the reference's result "
                                "objects, cc curves computed); rows equal to the resident-vector run"}
        for h in host:
            for a in h:
                if a is not None:
                    ctx.host_free(a)

    result = {
        "metric": "shifts*genome-bp/sec (whole node), hg38 max_shift=1000; HBM-BW fraction",
        "value": value,
        "unit": "shifts*bp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": (("BASELINE config 5 (stress): synthetic 10 Gbp genome F/R" if stress else
                          "BASELINE config 4: synthetic hg38-shaped F/R")
                         + ("+mappability" if with_m else "") + f" bit-vectors, {len(chroms)} chromosomes x "
                         f"{nsamples} genome(s), {total_bp / 1e9:.3f} Gbp total, max_shift={S}, read_len={L}, "
                         f"read density {args.density}/strand, "
                         + ("" if args.track == "synthetic" or not with_m else
                            "mappability runs and gaps sampled from the reference's test track hg19_36mer-test.bedGraph "
                            "(860 run edges per 64 Kbit, 23 % mappable), ")
                         + ("NCC+MSCC" if with_m else "NCC only")
                         + ("" if stress else
                            "; stands in for ENCFF000VPI.bam (configs 2-3), which is not available offline")),
            "mode": args.mode,
            "kernel_path": args.path,
            "window_only_hint": bool(hinted and dense),
            "deep_lists_hint": bool(hinted and deep),
            "run_edges_per_64kbit": (round(2 * 65536 * sum(v.n_runs for v in vecs) / max(sum(v.length for v in vecs), 1), 1)
                                     if with_m and vecs else None),
            "parallelism": (f"{nsamples} genome(s) cut into equal tile ranges over {world} GPU(s), one process per GPU; one all-reduce(sum) "
                            "of the per-chromosome rows on a second stream" if shard_tiles else
                            f"chromosome jobs of {nsamples} genome(s) LPT-sharded over {world} GPU(s), one process per "
                            "GPU; all-gather rows + all-reduce totals on a second stream")
                           + (" (1-rank RCCL group, collectives forced)" if args.force_collectives and world == 1 else ""),
            "inputs_resident_in_hbm": True,
            "workload_tag": workload_tag,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ctx.kernel_name(dom),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "avg_launch_ms": avg_ms,
            "launches": dom_n,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            # the reference's algorithm as dense integer work -- (S+1) N / 32 word-steps of >= 3 (NCC) / 9
            # (NCC+MSCC) lane-ops -- against the vector peak of 7.9e13 lane-ops/s.  The event formulation does not execute
            # that work (it enumerates ~3000 events per 64 Kbit instead), so this reads ABOVE 1: it says how far the
            # reformulation is from the dense algorithm's own roofline, not how busy the vector pipes are (see `issue`).
            "dense_lane_ops_per_launch": dense_lane_ops,
            "dense_equivalent_frac_of_valu_peak": (dense_lane_ops / (avg_ms * 1e-3) / 7.9e13) if avg_ms > 0 else None,
            "issue": issue_rate(ctx.kernel_name(dom), workload_tag),
        },
        "multi_gpu": multi_gpu,
        "repetitions": {"n": len(reps), "steps_each": args.steps, "ms_per_step_min_median_max":
                        [round(rep_ms[0], 5), round(rep_ms[len(rep_ms) // 2], 5), round(rep_ms[-1], 5)]},
        "pipe_utilisation": pipe_utilisation(ctx.kernel_name(dom), workload_tag),
        "build_id": ffi.build_id(),
        # HIP-event durations per step.  k_cc_events takes the sparse tiles (and the run-edge pairs of the mappable-length
        # pass); k_cc_sparse / k_autocorr_pairs+edges are the window kernels, which only see the tiles it flagged as dense
        # (none on this workload: their figure is an empty launch) -- or everything when max_shift > 1023
        "kernel_ms_per_step": kernel_ms_per_step,
        "value_end_to_end": end_to_end["value"] if end_to_end else None,
        "end_to_end": end_to_end,
        "end_to_end_calculator": calc_leg,
        "gen_seconds": round(t_gen, 2),
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores, quota = usable_cores()
        threads = args.cpu_threads or cores    # `-p <ncores>` = every core this process may use
        result["cpu_baseline"] = cpu_baseline(ctx, vecs, S, L, with_m, args.cpu_sample_mbp * 1e6, threads)
        result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
    else:
        result["cpu_baseline"] = None

    result["ingest"] = ingest_leg_isolated(args.ingest_reads) if (rank == 0 and world == 1 and not args.no_ingest) else None

    if rank == 0:
        print(json.dumps(result))
    ctx.close()
    if world > 1 or args.force_collectives:
        dist.destroy_process_group()


if __name__ == "__main__" and len(sys.argv) == 3 and sys.argv[1] == "--ingest-child":
    print(json.dumps(ingest_leg(int(sys.argv[2]))))
    sys.exit(0)

if __name__ == "__main__":
    main()
