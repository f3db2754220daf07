#!/usr/bin/env python3
"""paths/sec (train+score) of the KPRN hot path on MI355X -- BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic paths already resident in
HBM: ONE scoring forward (release/songPathRnn/eval/test_from_checkpoint.lua:109) with the parameters
as they stand, and ONE MyOptimizer:trainBatch (zeroPad, zeroGrad, forward, BCE, backward, Adam, zeroPad;
release/songPathRnn/model/optimizer/MyOptimizer.lua:177-221) over the same batch, i.e. the
"combined = N_paths / (t_train + t_score)" of SURVEY.md section 8d.  value = paths processed by
all ranks / max-over-ranks wall time of the K timed steps.

Workload = BASELINE.json configs[1] ("C2", reading A of "d=64"): path_len T=6, D=H=64
(d_type 16 | d_entity 32 | d_relation 16), 2-layer FastLSTM, fp32, KKBox vocabulary
(2 851 220 entities / 6 types / 9 relations, run_scripts/config.sh:24-26), 46 labels, LogSumExp
pool, Adam lr 1e-3.  Batches are bucketed by #paths-per-pair like the reference's files
(movie_data_format.py:301-314) and sized to ~--paths-per-step paths.

N>1: one process per GPU, pairs sharded across ranks, dense-gradient all-reduce + sparse entity-row
all-gather over RCCL (kprn_amd/dp.py); weak scaling (fixed --paths-per-step per GPU) or, with
--total-paths M, the M paths of the job split over ranks and steps (strong scaling).  Started as
`python bench.py --gpus N` (no WORLD_SIZE in the environment) it launches its N ranks itself through
torch.distributed.run on 127.0.0.1; started by torch.distributed.run it is one of the ranks.

Batch feed.  `value` is measured with the batches resident in HBM (ids uploaded, validated, indexed, planned
before the timed region).  The same K steps are then repeated with the STREAMING feed -- every step a batch that
arrives from (page-locked) host memory: upload + validation + occurrence index + identical-prefix plan on the
engine's feed stream, double-buffered under the previous step (kprn_batch_feed_async; the reference's
BatcherFileList.lua:53-96) -- and reported under "streaming"; `value_no_prefix_plan` repeats them with every
step of every path executed (the plan's saving depends on the share of left-padded paths in the data), and
"long_run" with enough steps for a >= 0.3 s timed region.

Extra objects on the JSON line: "roofline" (dominant kernel family, HIP events on the engine's
stream inside the timed region) and "cpu_baseline" (the float64 oracle with OpenMP on the host
cores, bounded sample, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS_F32_MFMA = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0          # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--paths-per-step", type=int, default=65536, help="paths per rank per step")
    ap.add_argument("--dims", default="A", choices=["A", "B", "shipped", "gru", "C4"],
                    help="A: D=H=64 (16/32/16); B: 64/64/64 -> D=H=192; shipped: run_scripts/config.sh as shipped (rnn, 50/100/50 -> D=200, H=250, L=1); "
                         "gru: config.sh's sizes with -rnnType gru (OneModel.lua:237-238); "
                         "C4: BASELINE configs[3] -- 20 M entities, 100 relations, d = 128 -> D = H = 384, L = 1, bf16 storage + bf16 MFMA "
                         "(--c4-fp32: the same shape in fp32)")
    ap.add_argument("--c4-fp32", action="store_true")
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--T", type=int, default=6)
    ap.add_argument("--entities", type=int, default=0, help="entity vocabulary (default: 2 851 220, KKBox; 20 000 000 for --dims C4)")
    ap.add_argument("--impl", default="auto", choices=["auto", "generic"])
    ap.add_argument("--entity-update", type=int, default=0, help="0 lazy-exact, 1 dense (as the reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--train-only", action="store_true")
    ap.add_argument("--score-only", action="store_true")
    ap.add_argument("--no-batch-sweep", action="store_true")
    ap.add_argument("--dp-unfused", action="store_true", help="data-parallel step: separate marking merge + row update instead of the union inside the row update (A/B)")
    ap.add_argument("--dp-score-first", action="store_true", help="data-parallel step: queue the scoring pass first (as the plain step does: it shares the chip with the training forward); default at world 1, where no collective needs hiding")
    ap.add_argument("--dp-score-under-gather", action="store_true", help="data-parallel step: queue the scoring pass behind the backward, while the all-gather is in flight; default at world > 1")
    ap.add_argument("--dp-score-split", type=float, default=-1.0, help="data-parallel step: this fraction of the scoring pass's tiles runs behind the backward, under the all-gather; the rest is queued first, beside the training forward (opt-in; default at world > 1: the whole pass under the gather)")
    ap.add_argument("--dp-torch-collectives", action="store_true", help="data-parallel step: the collectives through torch.distributed (hooks kprn_sparse_grad_pack / _merge) instead of the engine's own RCCL exchange")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact lines of the other BASELINE configs (each a short run of this script)")
    ap.add_argument("--cpu-baseline-quick", action="store_true", help="cpu_baseline: the model-only oracle sample only (no literal flavour, no torch-CPU point)")
    ap.add_argument("--reserve-cus", type=int, default=16, help="CUs the scoring pass leaves to the collective in data-parallel runs")
    ap.add_argument("--compute-dtype", type=int, default=0, choices=[0, 1, 2, 3],
                    help="0: fp32 MFMA (default, the headline).  2: f32x6 -- fp32 products formed exactly from bf16 pieces on the matrix "
                         "cores (forward only; same parity bars).  1: bf16 products (scoring only on the fused path)")
    ap.add_argument("--no-alt", action="store_true", help="skip the short second measurement with compute_dtype 2 (f32x6) that is reported under \"alt_f32x6\"")
    ap.add_argument("--no-score-overlap", action="store_true",
                    help="run the scoring pass on the main stream, strictly before the train step (default: on a second stream, sharing "
                         "the chip with the training forward of the same step -- neither depends on the other)")
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel step (pack / all-gather / merge) even at world size 1")
    ap.add_argument("--total-paths", type=int, default=0,
                    help="strong scaling: the K timed steps of all ranks together process this many paths (north_star: 1000000), "
                         "i.e. paths per rank per step = total / (gpus * steps); 0 = weak scaling with --paths-per-step per GPU")
    ap.add_argument("--batch-feed", default="both", choices=["resident", "streaming", "both"],
                    help="resident: batches in HBM before the timed region (this is `value`); streaming: every step's batch is uploaded, "
                         "validated, indexed and planned on the feed stream under the previous step; both: `value` resident + a second "
                         "region reported under \"streaming\"")
    ap.add_argument("--feed-build", default="host", choices=["host", "device"],
                    help="streaming feed: derive plan + index on host worker threads (GPU sees DMA only) or with kernels on a side stream")
    ap.add_argument("--feed-ahead", type=int, default=4, help="streaming feed: batches in flight ahead of the step being queued")
    ap.add_argument("--feed-threads", type=int, default=0, help="helper threads per batch of the host-built feed (0: library default)")
    ap.add_argument("--no-extra-regions", action="store_true", help="skip the value_no_prefix_plan / long_run regions")
    ap.add_argument("--workload", default="c2", choices=["c2", "c5"],
                    help="c2 (default): BASELINE configs[1], train + score at T = 6.  c5: configs[4] -- inference-only scoring of a path set with "
                         "variable path length <= 7, bucketed by identical T (3..7) as the reference pads per file, d = 64")
    ap.add_argument("--set-option", action="append", default=[], metavar="KEY=VALUE", help="kprn_set_option(KEY, VALUE) on the engine before the first step (A/B runs), repeatable")
    ap.add_argument("--uniform-tiles", action="store_true",
                    help="every path has T real steps (no left padding -> no identical-prefix skipping) and the batch is a whole number of 256 x 64-path "
                         "tiles: every workgroup of the fused kernels draws the same number of tile-steps.  The dominant kernel's roofline.frac on this "
                         "line is the kernel's own ceiling, free of the 18-vs-20 tile-step quantisation of the 4 / 6-step mix (DESIGN.md section 7-1)")
    ap.add_argument("--stream-total-paths", type=int, default=0,
                    help="after the headline region: this many paths in all through the STREAMING feed (every step's batch arrives from page-locked host memory "
                         "through the slot ring), reported under \"streaming_total\" with its ratio to the resident rate -- BASELINE configs[3] names 50 M paths")
    ap.add_argument("--no-strong-1m", action="store_true",
                    help="data-parallel runs without --total-paths also time the north_star's strong-scaling experiment (1 000 000 paths split over ranks "
                         "and steps, and the same total on rank 0 alone) and report it under \"strong_1M\"; this flag skips it")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rank wiring / timing / JSON check without a GPU: gloo backend, the step is a placeholder (CPU tests)")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, torch.distributed.run on
    127.0.0.1) and hand their exit code back.  Rank 0 prints the JSON line."""
    import subprocess
    if not a.dry_run:
        import torch
        n = torch.cuda.device_count()
        if n < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {n} GPU(s) visible")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


MIN_PATHS_PER_RANK_STEP = 32768


def strong_steps(total_paths, world, steps):
    """--total-paths (strong scaling: a FIXED total divided over ranks and steps): the step count is lowered until a rank's step holds
    >= 32 768 paths -- below ~8 k paths a step is the latency of one tile through four persistent kernels, which would
    measure that floor, not the scaling.  1 M paths on 8 GPUs -> 3 steps of 41 666 paths per rank instead of 20 steps of 6 250."""
    return max(1, min(steps, total_paths // (world * MIN_PATHS_PER_RANK_STEP)))


def replica_digests(vectors, world, device="cpu"):
    """Self-check of a data-parallel run: a 64-bit digest (blake2b) of each of this rank's `vectors` (numpy arrays), all-gathered over the
    default process group -> (per-rank hex digests, all ranks identical?).  Replicas hold the same bits by construction (rank-ordered sums:
    DESIGN.md section 4); a run whose replicas diverged exits with code 3."""
    import hashlib
    import torch
    import torch.distributed as dist
    digs = [int.from_bytes(hashlib.blake2b(np.ascontiguousarray(v).tobytes(), digest_size=8).digest(), "little", signed=True) for v in vectors]
    mine = torch.tensor(digs, dtype=torch.int64, device=device)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    allr = [[int(v) for v in t.cpu().tolist()] for t in allr]
    return [[f"{v & 0xffffffffffffffff:016x}" for v in r] for r in allr], bool(all(r == allr[0] for r in allr))


def dry_run(a):
    """No GPU, no engine: the ranks rendezvous over gloo, run K placeholder steps through the same barrier / max-over-ranks
    timing as the real run and rank 0 prints a JSON line marked dry_run.  Checks the launcher and the rank plumbing only."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if a.total_paths:
        a.steps = strong_steps(a.total_paths, world, a.steps)
    pps = a.paths_per_step if not a.total_paths else max(64, a.total_paths // (world * max(1, a.steps)))
    g = torch.ones(1024)   # the "replica": every rank applies the same all-reduced update
    cur_pps = [pps]
    def step():
        upd = torch.full((1024,), 1.0 + rank)
        if world > 1:
            dist.all_reduce(upd)
        g.add_(upd, alpha=1e-3)
        return cur_pps[0]
    # the bootstrap of the engine's own communicator, as a placeholder under the SAME watchdog and agreement as the real one (kprn_amd/dp.py
    # GpuAdapter.native_setup): a bootstrap that never returns on some rank makes every rank fall back, with the reason on the line
    from kprn_amd.dp import call_with_watchdog, first_reason
    def init():
        if os.environ.get("KPRN_DP_TEST_INIT_HANG") == "1" and rank == world - 1:
            time.sleep(3600)
    ok, why = call_with_watchdog(init, float(os.environ.get("KPRN_DP_INIT_TIMEOUT", "60")), "kprn_dp_init (dry-run placeholder)")
    okt = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if world > 1:
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    native = bool(int(okt.item()))
    dp_extra = {"exchange": "engine (placeholder)" if native else "torch.distributed collectives (fallback)",
                "fallback_reason": None if native else (first_reason(why) or "kprn_dp_init failed on another rank")}
    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    n = sum(step() for _ in range(a.steps))
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0, float(n), 1.0], dtype=torch.float64)
    tot = el.clone()
    if world > 1:
        mx = el.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        el[0] = mx[0]
    # the strong-scaling experiment beside a weak line (main(): strong_1m): the same total split over ranks and steps, then on rank 0 alone
    if world > 1 and not a.total_paths and not a.no_strong_1m:
        ks = strong_steps(STRONG_TOTAL, world, a.steps)
        cur_pps[0] = max(64, STRONG_TOTAL // (world * ks))
        dist.barrier()
        t1 = time.perf_counter()
        ns = sum(step() for _ in range(ks))
        dist.barrier()
        es = torch.tensor([time.perf_counter() - t1, float(ns)], dtype=torch.float64)
        mx = es.clone(); sm = es.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        k1 = strong_steps(STRONG_TOTAL, 1, a.steps)
        pps1 = max(64, STRONG_TOTAL // k1)
        n1 = None
        dist.barrier()
        if rank == 0:
            t2 = time.perf_counter()
            g1 = torch.ones(1024)
            for _ in range(k1):
                g1.add_(torch.full((1024,), 1.0), alpha=1e-3)
            e1 = max(time.perf_counter() - t2, 1e-9)
            n1 = {"steps": k1, "paths_per_step": pps1, "value": round(k1 * pps1 / e1, 1), "paths_counted": k1 * pps1}
        dist.barrier()
        v = float(sm[1]) / max(float(mx[0]), 1e-9)
        dp_extra["strong_1M"] = {"total_paths": STRONG_TOTAL, "steps": ks, "paths_per_rank_step": cur_pps[0], "value": round(v, 1), "unit": "paths/s",
                                 "paths_counted": int(sm[1]), "n1": n1,
                                 "speedup_vs_n1": round(v / n1["value"], 6) if n1 else None, "efficiency": round(v / n1["value"] / world, 6) if n1 else None}
        cur_pps[0] = pps
    if os.environ.get("KPRN_DRYRUN_DIVERGE") == "1" and rank == world - 1:   # (tests: a replica that went its own way must fail the run)
        g[7] += 1e-6
    per_rank, same = replica_digests((g.numpy(),), world)
    if rank == 0:
        print(json.dumps({"metric": "paths/sec (train+score) at path_len=6 d=64", "value": None, "unit": "paths/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * float(el[0]) / max(1, a.steps), 4),
                          "higher_is_better": True, "scaling": "strong" if a.total_paths else "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "none (dry run: placeholder steps, launcher and rank wiring only)", "dry_run": True,
                          "ranks_reporting": int(tot[2]), "paths_counted": int(tot[1]),
                          "dp": dict({"replica_digests": {"per_rank": per_rank}, "replicas_bit_identical": same}, **dp_extra),
                          "config": {"workload": "dry run", "paths_per_step_per_gpu": pps, "parallelism": f"dp{world}" if world > 1 else "single"}}))
    if world > 1:
        dist.destroy_process_group()
    if not same:
        sys.exit(3)


PEAK_TFLOPS_BF16_MFMA = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


def dims_of(a):
    if a.dims == "A":
        return 16, 32, 16, 64
    if a.dims == "C4":
        return 128, 128, 128, 384
    if a.dims in ("shipped", "gru"):
        return 50, 100, 50, 250
    return 64, 64, 64, 192


def family_work(name, N, T, D, H, L, C, F, nT, dt, de, dr, G=4, NT=None):
    """(bound, algorithmic work per launch) for a kernel family of the engine; None if unknown.
    G = rows of the recurrent weights per hidden unit (4 FastLSTM, 1 rnn).
    NT = (path, step) positions the fused kernels EXECUTE (kprn_batch_executed_steps: N*T less the identical leading
    steps run once per batch); a path's first executed step has no recurrent half.  Only executed flops are counted."""
    g = G * H
    if NT is None:
        NT = N * T
    if name in ("lstm_mc_fwd", "lstm_mc_fwd_train"):   # matrix-core forward: one launch per layer
        fl = 0
        for l in range(L):
            din = D if l == 0 else H
            fl += NT * 2 * g * din + (NT - N) * 2 * g * H
        return "mfma", (fl + N * 2 * H * C) / L
    if name in ("lstm_persist_bf16_score", "lstm_persist_bf16_train"):   # the whole layer in one launch (lstm_bf16_persist.hip); no recurrent half at t = 0
        return "mfma", N * 2 * g * (T * D + (T - 1) * H)
    if name == "lstm_persist_bf16_bwd":   # BPTT through the layer in one launch (lstm_bf16_bwd_persist.hip), HBM-bound: per (path, step) it reads the saved
        # gates (i, g, f, o: 8 B per hidden unit), c_t and c_{t-1} (2 + 2 B) and writes dA transposed (2 B per gate column)
        return "hbm", N * T * (H * 12 + g * 2)
    if name in ("lstm_step_fwd", "lstm_step_bf16", "rnn_step_fwd"):   # one launch per (layer, step): [x_t | h_{t-1}] [W_i | W_o]^T, no recurrent half at t = 0
        fl = 0
        for l in range(L):
            din = D if l == 0 else H
            fl += N * 2 * g * (T * din + (T - 1) * H)
        return "mfma", fl / (L * T)
    if name in ("lstm_layer_fwd", "rnn_layer_fwd", "gru_layer_fwd"):   # one launch per layer, all T steps (layer_f32_persist.hip); no recurrent half at t = 0
        fl = 0
        for l in range(L):
            din = D if l == 0 else H
            fl += N * 2 * g * (T * din + (T - 1) * H)
        return "mfma", fl / L
    if name in ("lstm_layer_bwd", "rnn_layer_bwd", "gru_layer_bwd"):   # BPTT of one layer in one launch: the recurrent products dh_{t-1} = dA_t W_o2g of steps T-1 .. 1 (gru: d(r h') = d pre_n c_h2h and [d pre_r | d pre_z] o2g: g = 3H)
        return "mfma", N * 2 * g * (T - 1) * H
    if name in ("lstm_fused_fwd", "lstm_fused_fwd_train"):
        fl = 0
        for l in range(L):
            din = D if l == 0 else H
            fl += NT * 2 * g * din + (NT - N) * 2 * g * H
        return "mfma", fl + N * 2 * H * C
    if name == "lstm_fused_bwd":
        fl = 0
        for l in range(L):
            din = D if l == 0 else H
            fl += 2 * (NT * 2 * g * din + (NT - N) * 2 * g * H)   # dW and [dx | dh]
        return "mfma", fl / L   # one launch per layer
    tbl = {
        "gemm_i2g_fwd": 2 * T * N * g * D, "gemm_o2g_fwd": 2 * N * g * H, "gemm_head_fwd": 2 * N * C * H,
        "gemm_o2g_bwd_dh": 2 * N * g * H, "gemm_o2g_bwd_dw": 2 * (T - 1) * N * g * H,
        "gemm_i2g_bwd_dw": 2 * T * N * g * D, "gemm_i2g_bwd_dx": 2 * T * N * g * D,
        # configs[3], small tables (lstm_bf16.hip k_onehot_T): dx for the entity slice only; ONE dW product over [x_e^T | S^T (128 one-hot rows) | h^T]
        "gemm_i2g_bwd_dx_e": 2 * T * N * g * de, "gemm_bwd_dw_merged": 2 * T * N * g * (de + 128 + H),
    }
    if name in tbl:
        return "mfma", tbl[name]
    if name == "embed_gather":
        return "hbm", N * T * ((nT * dt + de + dr) * 4 + F * 4)
    if name == "embed_scatter":   # the type / relation tables' gradients
        return "hbm", N * T * ((dt + dr) * 4 + F * 4)
    if name == "entity_grad":
        return "hbm", N * T * (de * 4 + 8)
    return None


def pmc_traffic(name, paths_per_step):
    """HBM bytes per launch of kernel family `name` from the rocprofv3 PMC passes of this very command
    (scripts/gpu_profile.sh -> profiles/<round>/pmc_summary_<tag>.json: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate
    passes).  bench.py cannot collect counters on itself; None when no summary for this workload size is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("paths_per_step") != paths_per_step:
            continue
        k = d.get("kernels", {}).get(name)
        if k and "hbm_bytes_per_launch" in k:
            best = {"hbm_bytes_per_launch": round(k["hbm_bytes_per_launch"]), "source": os.path.relpath(f, ROOT)}
    return best


def cpu_baseline(a, T, dt, de, dr, H, L, seconds):
    """float64 oracle (CPU restatement of the reference, kind "port") on the host cores:
    model-only train pass (forward + BCE + backward, no dense optimiser sweep) + scoring pass over
    the same bounded sample, like the GPU step.  A reported baseline, not the target."""
    from oracle.oracle import Oracle, make_cfg
    from kprn_amd import synth
    cores = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    Ve = 100000  # the entity table size does not affect the model-only flavour; keep RAM small
    cfg = make_cfg(Vt=6, Ve=Ve, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L)
    orc = Oracle(cfg, np.float64)
    theta = orc.init_params(1, 0.1)
    P = 2
    pairs = 2048
    idx, labels = synth.make_paths(pairs, P, T, Ve=Ve, seed=9)
    orc.forward_backward(theta, idx[:256], labels[:256])   # (warm-up: the OpenMP team, first touch of the buffers)
    t0 = time.perf_counter()
    orc.forward_backward(theta, idx, labels)
    orc.forward(theta, idx)
    dt0 = time.perf_counter() - t0
    rate = pairs * P / max(dt0, 1e-6)
    pairs = int(max(256, min(2000000, rate * seconds / P)))   # the timed sample: ~`seconds` of host work
    idx, labels = synth.make_paths(pairs, P, T, Ve=Ve, seed=10)
    t0 = time.perf_counter()
    orc.forward_backward(theta, idx, labels)
    t1 = time.perf_counter()
    orc.forward(theta, idx)
    t2 = time.perf_counter()
    n = pairs * P
    out = {"value": n / (t2 - t0), "unit": "paths/s", "cores": cores, "kind": "port",
           "sample": f"{n} synthetic paths (P={P}, T={T}, D=H={H}, L={L}): float64 oracle forward+BCE+backward then scoring "
                     f"forward, OpenMP over pairs; model-only flavour (no dense Adam sweep over the entity table)",
           "train_paths_per_s": n / (t1 - t0), "score_paths_per_s": n / (t2 - t1)}
    if not a.cpu_baseline_quick:
        try:
            out["literal"] = cpu_baseline_literal(a, T, dt, de, dr, H, L, max(5.0, seconds / 2), cores)
        except MemoryError as e:   # (a small host: the flat float64 vector + Adam state of the KKBox table is ~3 GB)
            out["literal"] = {"error": f"MemoryError: {e}"}
        out["torch_cpu_lstm"] = cpu_baseline_torch(T, dt, de, dr, H, L, max(5.0, seconds / 2), cores)
    return out


def cpu_baseline_literal(a, T, dt, de, dr, H, L, seconds, cores):
    """SURVEY 8d flavour (i): the reference's step AS WRITTEN at its own minibatch -- MyOptimizer.lua:186 zeroGradParameters over the whole
    flat vector, forward + BCE + backward, :218 optim.adam over every parameter including all of entity_emb (Ve rows, here the
    KKBox size) -- in the float64 oracle, OpenMP.  What the dense passes cost the reference per minibatch of config.sh:38's 128 pairs."""
    from oracle.oracle import Oracle, make_cfg, make_opt
    from kprn_amd import synth
    Ve = a.entities if a.entities > 0 else 2851220
    cfg = make_cfg(Vt=6, Ve=Ve, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L)
    orc = Oracle(cfg, np.float64)
    theta = orc.init_params(1, 0.1)
    st = orc.new_state()
    opt = make_opt(method=1, lr=1e-3)
    pairs, P = 128, 2
    idx, labels = synth.make_paths(pairs, P, T, Ve=Ve, seed=11)
    orc.train_step(theta, st, opt, idx, labels)   # (first touch of 4 x 0.73 GB)
    steps, t0 = 0, time.perf_counter()
    while True:
        orc.train_step(theta, st, opt, idx, labels)
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds or steps >= 2000:
            break
    return {"value": steps * pairs * P / el, "unit": "paths/s", "cores": cores, "kind": "port", "ms_per_step": round(1e3 * el / steps, 3),
            "steps": steps, "params": int(orc.n),
            "sample": f"{steps} trainBatch steps of {pairs} pairs x {P} paths (config.sh:38 minibatch), T={T}, D=H={H}, L={L}, Ve={Ve}: dense "
                      f"zeroGrad + forward + BCE + backward + dense optim.adam over all {orc.n} float64 parameters per step"}


def cpu_baseline_torch(T, dt, de, dr, H, L, seconds, cores):
    """Second, independently implemented CPU point: torch-CPU float64 -- torch.nn.LSTM (its own BLAS-backed kernel; FastLSTM's
    gate order differs only by a row permutation of the weights, irrelevant to the cost), embedding gathers, Linear(H, 46), LSE pool,
    BCE, autograd backward; then a scoring forward -- model-only flavour on a bounded sample, like the oracle line above."""
    import torch
    # (torch's intra-op pool over all 256 hardware threads of the GPU box ran this model at 98 paths/s: the LSTM's small GEMMs drown in
    # fork-join overhead; 16 threads is where it stops scaling on this shape)
    cores = min(cores, 16)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(5)
    D, C, Ve = dt + de + dr, 46, 100000
    emb = [torch.nn.Embedding(6, dt), torch.nn.Embedding(Ve, de), torch.nn.Embedding(9, dr)]
    lstm = torch.nn.LSTM(D, H, num_layers=L, batch_first=True)
    head = torch.nn.Linear(H, C)
    mods = torch.nn.ModuleList(emb + [lstm, head]).double()
    def fwd(idx, P):
        x = torch.cat([emb[k](idx[..., k]) for k in range(3)], dim=-1)      # [N, T, D]
        h, _ = lstm(x)
        s = head(h[:, -1, :]).view(-1, P, C)
        return torch.sigmoid(torch.logsumexp(s, dim=1))[:, 0]
    def sample(pairs, P):
        idx = torch.stack([torch.randint(0, 6, (pairs * P, T), generator=g), torch.randint(0, Ve, (pairs * P, T), generator=g),
                           torch.randint(0, 9, (pairs * P, T), generator=g)], dim=-1)
        return idx, torch.randint(0, 2, (pairs,), generator=g).double()
    def one(idx, lab, P):
        mods.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        loss = torch.nn.functional.binary_cross_entropy(fwd(idx, P), lab)
        loss.backward()
        t1 = time.perf_counter()
        with torch.no_grad():
            fwd(idx, P)
        return t1 - t0, time.perf_counter() - t1
    P = 2
    idx, lab = sample(4096, P)    # pairs per autograd pass (bounded activation memory)
    one(idx[:512], lab[:256], P)  # (warm-up: thread pool, allocator)
    n, tt, ts = 0, 0.0, 0.0
    while tt + ts < seconds:
        x, y = one(idx, lab, P)
        tt += x; ts += y; n += idx.shape[0]
    return {"value": n / (tt + ts), "unit": "paths/s", "cores": cores, "kind": "independent (torch-CPU float64 nn.LSTM)",
            "train_paths_per_s": n / tt, "score_paths_per_s": n / ts,
            "sample": f"{n} synthetic paths (P={P}, T={T}, D={D}, H={H}, L={L}) in passes of {idx.shape[0]} paths: forward + BCE + autograd backward, then a "
                      f"no-grad scoring forward; torch {torch.__version__}, {cores} threads"}


def dropin_minibatch(eng, opt, T, F, Vt, Ve, Vr, nT, seconds=0.4):
    """The calls bindings/kprn.lua makes in place of MyOptimizer:trainBatch (MyOptimizer.lua:177-221) and test_from_checkpoint.lua:109, at the
    reference's own sizes: kprn_train_step from HOST buffers with 128 pairs (run_scripts/config.sh:38), the loss returned to the host every
    step; then kprn_forward from host buffers with 512 pairs (test_from_checkpoint.lua:49), the probabilities returned.  A minibatch comes from
    one bucket file (constant P: movie_data_format.py:311-314); P is drawn per minibatch from the fixture's distribution (SURVEY 8d:
    min(Geom(0.57), 28), mean 1.75).  Every call returns its result to the host: kprn_forward after the pass, kprn_train_step as soon as the loss stage
    has run (the backward and the update follow in stream order while the caller prepares its next minibatch; the region ends with a drain)."""
    from kprn_amd import synth
    rng = np.random.default_rng(99)
    def pool(pairs, n, seed0):
        out = []
        for i in range(n):
            P = int(min(rng.geometric(0.57), 28))
            idx, labels = synth.make_paths(pairs, P, T, F=F, Vt=Vt, Ve=Ve, Vr=Vr, num_types=nT, seed=seed0 + i)
            out.append((np.ascontiguousarray(idx, np.int32), np.ascontiguousarray(labels, np.float32)))
        return out
    train, score = pool(128, 32, 9100), pool(512, 16, 9300)
    def run(fn, items, seconds):
        # warm-up: every minibatch once through each of the two alternating engine-owned slots, so that both have grown to the largest P before the clock starts
        # (BatcherFileList.lua:53-60 preallocates its GPU tensors for the largest file the same way; rounds 3-4 warmed six minibatches only and timed the slots'
        #  re-allocations -- page-locked staging included -- with the steps)
        for it in items + items[:1] + items:
            fn(it)
        eng.sync()
        k, n, t0 = 0, 0, time.perf_counter()
        while True:
            idx, _ = items[k % len(items)]
            fn(items[k % len(items)])
            n += idx.shape[0] * idx.shape[1]
            k += 1
            if k % 8 == 0 and time.perf_counter() - t0 >= seconds:
                break
        eng.sync()   # (kprn_train_step returns with the loss; the last step's backward and update are still part of the region)
        el = time.perf_counter() - t0
        return {"steps": k, "steps_per_s": round(k / el, 1), "paths_per_s": round(n / el, 1), "ms_per_step": round(1e3 * el / k, 4),
                "mean_paths_per_step": round(n / k, 1)}
    losses = []
    tr = run(lambda it: losses.append(eng.train_step_host(it[0], it[1], opt)), train, seconds)
    sc = run(lambda it: eng.forward_host(it[0], 1, want_all=False), score, seconds)   # (the scorer reads preds[i] only: test_from_checkpoint.lua:82,109)
    assert np.all(np.isfinite(losses))
    return {"train_128_pairs": tr, "score_512_pairs": sc,
            "what": "kprn_train_step / kprn_forward from host buffers at the reference's minibatch sizes (config.sh:38, test_from_checkpoint.lua:49), "
                    "P per minibatch ~ min(Geom(0.57), 28); loss / probabilities returned to the host every call"}


def other_configs():
    """compact {value, ms_per_step, roofline} of configs[4] (inference buckets), "d = 64" reading B, run_scripts/config.sh as shipped and
    configs[3] (20 M entities, bf16) -- `python bench.py <flags>` each, short, no CPU leg, no extra regions"""
    import subprocess
    # step counts chosen for a timed region of >= 0.3 s each (C5 0.35 ms, dims B 20 ms, shipped 5.5 ms, configs[3] 5-7 ms per step)
    runs = [("C5_inference_T3to7", ["--workload", "c5", "--steps", "1000", "--warmup", "10"]),
            ("dimsB_D192_H192_L2", ["--dims", "B", "--steps", "18", "--warmup", "2"]),
            ("shipped_rnn_D200_H250", ["--dims", "shipped", "--steps", "72", "--warmup", "3"]),
            ("gru_D200_H250", ["--dims", "gru", "--steps", "30", "--warmup", "3"]),
            ("C4_20M_entities_d128_bf16", ["--dims", "C4", "--steps", "75", "--warmup", "3", "--stream-total-paths", "50000000"])]
    out = {}
    for name, flags in runs:
        cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-alt", "--no-extra-regions", "--no-other-configs", "--no-batch-sweep", "--batch-feed", "resident"] + flags
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1])
            rf = d.get("roofline") or {}
            out[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d.get("dtype"), "steps": d.get("steps"),
                         "timed_region_s": round(d["ms_per_step"] * d.get("steps", 0) / 1e3, 3), "mfma_frac_end_to_end": d.get("mfma_frac_end_to_end"),
                         "mfma_frac_end_to_end_executed": d.get("mfma_frac_end_to_end_executed"),   # (issued MFMA flops only: the small-table identity removes products)
                         "roofline": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")},
                         "wall_s": round(time.perf_counter() - t0, 1), "flags": " ".join(flags)}
            if d.get("streaming_total"):   # configs[3] at the size the config names: 50 M paths streamed from host memory
                out[name]["streaming_50M"] = {k: d["streaming_total"].get(k) for k in ("value", "unit", "ms_per_step", "steps", "total_paths", "seconds", "ratio_to_resident")}
            oth = rf.get("other_timed_families")
            if oth:
                out[name]["roofline"]["other_timed_families"] = {k: {q: v.get(q) for q in ("bound", "frac", "avg_launch_ms")} for k, v in oth.items() if v}
        except Exception as e:   # a failing side run must not take the headline line with it
            out[name] = {"error": repr(e)[:300], "flags": " ".join(flags)}
    return out


def run_c5(a):
    """BASELINE configs[4]: inference-only scoring throughput, variable path length <= 7 with bucketed batching (the reference pads
    every path of a file to that file's T -- movie_data_format.py:250-254 -- so a bucket = the paths of one T; pads are ordinary
    steps for FastLSTM), d = 64 (16 / 32 / 16, H = 64, L = 2), KKBox-size vocabulary, LSE pool.  A step = one scoring
    forward (test_from_checkpoint.lua:109) over one bucket batch of ~--paths-per-step paths; buckets T = 3..7 in turn."""
    import torch
    assert a.gpus == 1, "the c5 line is a single-GPU measurement"
    torch.cuda.set_device(0)
    from kprn_amd import _ffi, synth
    dt_, de_, dr_, H, L, C, F = 16, 32, 16, 64, 2, 46, 3
    Vt, Ve, Vr = 6, (a.entities if a.entities > 0 else 2851220), 9
    stream = torch.cuda.current_stream().cuda_stream
    eng = _ffi.Engine(Vt, Ve, Vr, dt_, de_, dr_, H, L, F=F, C_=C, reducer=2, stream=stream, compute_dtype=a.compute_dtype)
    eng.set_option("impl", a.impl)
    Ts = [3, 4, 5, 6, 7]
    host, batches = [], []
    for i, T in enumerate(Ts):
        P = [2, 1, 3, 4, 2][i]
        idx, _ = synth.make_paths(max(1, a.paths_per_step // P), P, T, F=F, Vt=Vt, Ve=Ve, Vr=Vr, seed=777 + i)
        hi = eng.host_array(idx.shape, np.int32)
        hi[...] = idx
        host.append(hi)
        batches.append(eng.batch(hi))
    paths_of = [b.n_paths for b in batches]
    exec_of = [b.executed_steps for b in batches]
    D = dt_ + de_ + dr_
    def flops(i):
        return family_work("lstm_fused_fwd", paths_of[i], Ts[i], D, H, L, C, F, 1, dt_, de_, dr_, 4, exec_of[i])[1]
    def region(step, first, k):
        torch.cuda.synchronize(); eng.sync()
        t0 = time.perf_counter()
        n = sum(step(first + i) for i in range(k))
        eng.sync(); torch.cuda.synchronize()
        return time.perf_counter() - t0, n
    def step_res(i):
        eng.forward_async(batches[i % len(batches)], 1)
        return paths_of[i % len(batches)]
    NS = max(1, a.feed_ahead) + 1
    slots = [_ffi.Batch.reserve(eng, max(b.B for b in batches), max(paths_of), 7, F, with_labels=False) for _ in range(NS)]
    fed = [-1]
    def step_str(i):
        if fed[0] < i - 1 or fed[0] >= i + NS - 1:
            fed[0] = i - 1
        while fed[0] < i + NS - 1:
            j = fed[0] + 1
            slots[j % NS] = eng.feed(host[j % len(host)], None, slot=slots[j % NS])
            fed[0] = j
        eng.forward_async(slots[i % NS], 1)
        return paths_of[i % len(host)]
    for i in range(a.warmup):
        step_res(i)
    eng.sync()
    eng.profile_reset(); eng.set_option("profile_filter", "lstm_fused_fwd" if a.impl == "auto" else ""); eng.profile(not a.no_kernel_events)
    el, n = region(step_res, a.warmup, a.steps)
    eng.profile(False)
    fams = eng.profile_get()
    for i in range(3):
        step_str(i)
    eng.sync()
    el_s, n_s = region(step_str, 3, a.steps)
    roofline = None
    if "lstm_fused_fwd" in fams:
        ms, launches = fams["lstm_fused_fwd"]
        work = sum(flops((a.warmup + i) % len(batches)) for i in range(a.steps))
        roofline = {"kernel": "lstm_fused_fwd", "bound": "mfma", "achieved": round(work / (ms * 1e-3) / 1e12, 3), "peak": PEAK_TFLOPS_F32_MFMA,
                    "unit": "TFLOP/s", "frac": round(work / (ms * 1e-3) / 1e12 / PEAK_TFLOPS_F32_MFMA, 4), "traffic": None,
                    "avg_launch_ms": round(ms / max(launches, 1), 5), "launches": launches}
    cpu = None
    if not a.no_cpu_baseline:
        from oracle.oracle import Oracle, make_cfg
        cores = len(os.sched_getaffinity(0))
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        orc = Oracle(make_cfg(Vt=6, Ve=100000, Vr=9, dt=dt_, de=de_, dr=dr_, H=H, L=L), np.float64)
        theta = orc.init_params(1, 0.1)
        tot, t0 = 0, time.perf_counter()
        for T in Ts:
            idx, _ = synth.make_paths(8000, 2, T, Ve=100000, seed=5 + T)
            orc.forward(theta, idx)
            tot += 16000
        cpu = {"value": tot / (time.perf_counter() - t0), "unit": "paths/s", "cores": cores, "kind": "port",
               "sample": f"{tot} synthetic paths, 16000 per bucket T = 3..7: float64 oracle scoring forward, OpenMP over pairs"}
    print(json.dumps({
        "metric": "paths/sec (score only) variable path_len<=7 bucketed, d=64", "value": round(n / el, 1), "unit": "paths/s", "n_gpus": 1,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * el / a.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {0: "f32", 1: "bf16", 2: "f32x6", 3: "f32x3"}[a.compute_dtype], "data": "synthetic",
        "config": {"workload": f"C5 (BASELINE configs[4]) MovieLens-KG-shaped synthetic: inference only, buckets T = 3..7, D = H = 64 (16/32/16), L = 2, "
                               f"fp32, Ve = {Ve}, C = 46, LSE pool; one scoring forward per bucket batch", "paths_per_step": a.paths_per_step,
                   "buckets_T": Ts, "impl": a.impl, "batch_feed": "resident"},
        "executed_step_fraction": round(sum(exec_of) / float(sum(p * t for p, t in zip(paths_of, Ts))), 4),
        "streaming": {"value": round(n_s / el_s, 1), "unit": "paths/s", "ms_per_step": round(1e3 * el_s / a.steps, 4),
                      "ratio_to_resident": round((n_s / el_s) / (n / el), 4)},
        "roofline": roofline, "cpu_baseline": cpu}))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    if a.dry_run:
        return dry_run(a)
    if a.workload == "c5":
        return run_c5(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: kprn_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1 or a.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    requested_steps = a.steps
    if a.total_paths:
        a.steps = strong_steps(a.total_paths, world, a.steps)
        a.paths_per_step = max(64, a.total_paths // (world * max(1, a.steps)))
    if a.uniform_tiles:
        a.paths_per_step = max(256 * 64, (a.paths_per_step // (256 * 64)) * 256 * 64)

    from kprn_amd import _ffi, synth, dp
    dt_, de_, dr_, H = dims_of(a)
    shipped = a.dims == "shipped"
    gru = a.dims == "gru"
    c4 = a.dims == "C4"
    if c4 and a.compute_dtype == 0 and not a.c4_fp32:
        a.compute_dtype = 1
    if a.entities <= 0:
        a.entities = 20_000_000 if c4 else 2851220
    D, L, T, C, F, nT = dt_ + de_ + dr_, (1 if (shipped or c4 or gru) else a.layers), a.T, 46, 3, 1
    G = 1 if shipped else (3 if gru else 4)   # rows of recurrent weights per hidden unit (gru: r, z and the candidate)
    Vt, Ve, Vr = 6, a.entities, (100 if c4 else 9)
    stream = torch.cuda.current_stream().cuda_stream
    eng = _ffi.Engine(Vt, Ve, Vr, dt_, de_, dr_, H, L, F=F, num_types=nT, C_=C, reducer=2, device_id=local_rank,
                      rank=rank, world=world, param_init=0.1, seed=12345, stream=stream,
                      rnn_type=1 if shipped else (2 if gru else 0), use_relu=1, rnn_init=1 if shipped else 0, compute_dtype=a.compute_dtype)
    eng.set_option("impl", a.impl)
    eng.set_option("score_overlap", "0" if a.no_score_overlap else "1")
    eng.set_option("feed_build", a.feed_build)
    if a.feed_threads > 0:
        eng.set_option("feed_threads", str(a.feed_threads))
    for kv in a.set_option:
        k_, _, v_ = kv.partition("=")
        eng.set_option(k_, v_)
    opt = _ffi.make_opt(method=1, lr=1e-3, entity_update=a.entity_update)

    # bucketed batches (constant P per batch, like the reference's train.txt.<P>.torch files)
    Ps = [1, 2, 4, 8] if a.uniform_tiles else [1, 2, 3, 4, 5, 8]   # (uniform tiles: P divides the batch, so every batch is exactly paths_per_step paths)
    host = []      # the path set in host memory (page-locked: what a loader hands the feed)
    batches = []   # ... and resident in HBM
    for i, P in enumerate(Ps):
        pairs = max(1, a.paths_per_step // P)
        idx, labels = synth.make_paths(pairs, P, T, F=F, Vt=Vt, Ve=Ve, Vr=Vr, num_types=nT, seed=12345 + 97 * i + 7919 * rank,
                                       real_len=(T if a.uniform_tiles else None))
        hi, hl = eng.host_array(idx.shape, np.int32), eng.host_array(labels.shape, np.float32)
        hi[...] = idx
        hl[...] = labels
        host.append((hi, hl))
        batches.append(eng.batch(hi, hl))
    paths_of = [b.n_paths for b in batches]
    exec_of = [b.executed_steps for b in batches]

    dpx = None
    if world > 1 or a.force_dp:
        if a.dp_torch_collectives or a.dp_unfused:
            os.environ["KPRN_DP_NATIVE"] = "0"
        dpx = dp.DataParallel(dp.GpuAdapter(eng, dev, fused_update=not a.dp_unfused))
        # where the scoring pass goes: first (beside the training forward, as in the plain step) when no collective needs hiding, else behind
        # the backward while the all-gather is in flight (the engine's exchange then runs the collective on a stream of its own)
        dp_score_first = a.dp_score_first or (world == 1 and not a.dp_score_under_gather and a.dp_score_split <= 0)
        # the scoring pass split around the collective: most of it shares the chip with the training forward (as in the plain step), the
        # rest gives the all-gather something to hide under
        dp_split = 0.0
        # (measured at world 1, profiles/r04: plain 1.461 ms, pass first 1.509, all of it under the gather 1.672, split 0.5 / 0.3: 1.612 / 1.624 --
        #  work placed under the gather runs alone on 240 CUs in whole rounds of tiles, so the split buys 0.05 ms; opt-in)
        if not dp_score_first and not a.dp_score_under_gather and a.compute_dtype == 0 and a.dp_score_split > 0:
            dp_split = a.dp_score_split
        if 0.0 < dp_split < 1.0:
            eng.set_option("score_split", str(dp_split))
        if dpx.native and not dp_score_first:
            eng.set_option("dp_comm_stream", "1")
        # packing capacity = largest distinct-row count of any batch on any rank (known from the batch index)
        dpx.set_capacity(max(b.n_uniq for b in batches))

    # A step = one scoring forward + one trainBatch over the same batch.  The scoring pass uses the parameters as they are
    # BEFORE this step's update ("score the batch, then learn from it"), so it does not depend on the step's gradient
    # exchange: in data-parallel runs it is enqueued while the entity-row all-gather is in flight (kprn_amd/dp.py), with
    # a few CUs left free for the collective's copy kernels.
    if dpx is not None and not dp_score_first:   # (a pass queued first shares the chip with the training forward, not with a collective)
        eng.set_option("reserve_cus", str(a.reserve_cus))

    def run_batch(b):
        score = (lambda: eng.forward_async(b, 1)) if not a.train_only else None
        if a.score_only:
            eng.forward_async(b, 1)
        elif dpx is not None:
            # The scoring pass scores with the pre-update parameters, so it is independent of the exchange and the update waits for it.  It is
            # queued FIRST (beside the training forward, exactly as in the plain step) when no collective needs hiding (world 1), else behind
            # the backward while the all-gather is in flight.  profiles/r03 + DESIGN.md section 4 have the measured timelines of both.
            if score and dp_score_first:
                score()
                dpx.train_step(b, opt, 1, overlap=None)
            elif score and 0.0 < dp_split < 1.0:
                score()                                                       # first part: beside the training forward
                dpx.train_step(b, opt, 1, overlap=eng.forward_async_rest)     # the rest: under the all-gather
            else:
                dpx.train_step(b, opt, 1, overlap=score)
        else:
            if score:
                score()
            eng.train_step(b, opt, 1, want_loss=False)

    def step_resident(pool):
        def step(i):
            run_batch(pool[i % len(pool)])
            return paths_of[i % len(pool)]
        return step

    # streaming feed: AHEAD + 1 slots; the batch of step i + AHEAD is handed to the feed right before step i is queued.  Host
    # worker threads validate it and derive its identical-prefix plan and occurrence index into page-locked staging, and its
    # uploads (DMA: no CUs) start behind the slot's last reader (step i - 1); queueing step i + AHEAD waits for the worker, never for
    # the device (BatcherFileList.lua:53-96's double buffer, a few slots deeper)
    AHEAD = max(1, a.feed_ahead)
    NS = AHEAD + 1
    # slots sized once for the largest bucket (BatcherFileList.lua:53-60 preallocates its GPU tensors the same way)
    slots = ([_ffi.Batch.reserve(eng, max(b.B for b in batches), max(paths_of), T, F) for _ in range(NS)]
             if (a.batch_feed != "resident" or a.stream_total_paths > 0) else [None] * NS)
    fed_upto = [-1]
    host_t = {"feed": [], "run": []}   # host-side seconds per step spent queueing the feed / the step (KPRN_BENCH_HOST_TIMING=1 prints them)
    def step_streaming(i):
        if fed_upto[0] < i - 1 or fed_upto[0] >= i + AHEAD:   # (a region starts: restart the ring at i)
            fed_upto[0] = i - 1
        t0 = time.perf_counter()
        while fed_upto[0] < i + AHEAD:
            j = fed_upto[0] + 1
            slots[j % NS] = eng.feed(*host[j % len(host)], slot=slots[j % NS])
            fed_upto[0] = j
        t1 = time.perf_counter()
        run_batch(slots[i % NS])
        host_t["feed"].append(t1 - t0)
        host_t["run"].append(time.perf_counter() - t1)
        return paths_of[i % len(host)]

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or a.force_dp:
            dist.barrier()
        torch.cuda.synchronize()

    last_local = [0.0]   # this rank's own seconds of the last timed region (before the max over ranks)

    def timed_region(step, first, k):
        """exactly k steps between barriers; returns (max-over-ranks seconds, paths of all ranks)"""
        barrier()
        t0 = time.perf_counter()
        n = 0
        for i in range(k):
            n += step(first + i)
        barrier()
        el = time.perf_counter() - t0
        last_local[0] = el
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tn = torch.tensor([n], dtype=torch.float64, device=dev)
            dist.all_reduce(tn, op=dist.ReduceOp.SUM)
            return float(tt.item()), float(tn.item())
        return el, float(n)

    main_streaming = a.batch_feed == "streaming"
    step = step_streaming if main_streaming else step_resident(batches)

    # Warm-up steps run with HIP events around EVERY kernel family (untimed): that gives the per-family table and names the
    # dominant family.  Inside the timed region only the dominant family keeps its events (an event pair costs ~4 us of
    # stream time; 13 pairs per step were 6 % of the step), which is what `roofline` is computed from.
    prof = not a.no_kernel_events
    try:
        for i in range(a.warmup):
            if i == min(1, a.warmup - 1):  # (the very first step carries one-time costs: code load, allocations)
                eng.sync()
                eng.profile_reset()
                eng.profile(prof)
            step(i)
    except dp.DpHang as ex:
        # the engine's own collective never completed on the device: nothing on this stream can be waited for any more.  Say so and leave
        # (exit code 4) instead of hanging in the next synchronisation.
        if rank == 0:
            emit_last({"metric": "paths/sec (train+score) at path_len=6 d=64", "value": None, "unit": "paths/s", "n_gpus": world, "steps": 0,
                       "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                       "data": "synthetic", "config": {"workload": "aborted in the warm-up steps", "parallelism": f"dp{world}"},
                       "dp": {"world": world, "fallback_reason": str(ex), "aborted": True}}, hard_exit=True, rc=4)
        os._exit(4)
    eng.sync()
    fams_warm = eng.profile_get() if prof else {}
    # dominant family = the one that carries the most algorithmic work per step among the families whose work is known
    # (falls back to the largest time share).  By time alone the scoring forward would win whenever it shares the chip with
    # the training forward on its second stream: its launch duration then measures waiting, not work.
    dominant = ""
    if fams_warm:
        wsteps = max(1, a.warmup - min(1, a.warmup - 1))
        def work_per_step(name):
            fw = family_work(name, paths_of[0], T, D, H, L, C, F, nT, dt_, de_, dr_, G, exec_of[0])
            return None if fw is None else fw[1] * fams_warm[name][1] / wsteps
        known = {n: work_per_step(n) for n in fams_warm if work_per_step(n) is not None and family_work(n, 1, T, D, H, L, C, F, nT, dt_, de_, dr_, G)[0] == "mfma"}
        dominant = max(known, key=known.get) if known else max(fams_warm.items(), key=lambda kv: kv[1][0])[0]
    eng.profile_reset()
    if dominant.startswith("lstm_persist_bf16"):   # scoring and training launches of the persistent bf16 layer kernel: both keep their events
        dominant = "lstm_persist_bf16"
    eng.set_option("profile_filter", dominant)
    elapsed, npaths_total = timed_region(step, a.warmup, a.steps)
    elapsed_local = last_local[0]
    eng.profile(False)
    fams = eng.profile_get() if prof else {}
    value = npaths_total / elapsed

    # ---- further regions (not the headline; N = 1 only, no kernel events)
    def region_line(el, n, k):
        return {"value": round(n / el, 1), "unit": "paths/s", "ms_per_step": round(1e3 * el / k, 4), "steps": k}
    extras = {}
    plain = world == 1 and not a.force_dp and not a.no_extra_regions
    if plain and a.batch_feed == "both":
        # the same steps with the streaming feed (its own warm-up: the ring of slots fills, their staging is allocated; at least 60 steps,
        # so that the pipeline's start-up transient does not decide the figure)
        k_s = max(a.steps, 60)
        for i in range(12):
            step_streaming(i)
        eng.sync()
        el, n = timed_region(step_streaming, 12, k_s)
        if os.environ.get("KPRN_BENCH_STREAM_PROF"):   # (diagnosis: per-family times of a few streaming steps)
            eng.profile_reset(); eng.set_option("profile_filter", ""); eng.profile(True)
            timed_region(step_streaming, 12 + k_s, 12)
            eng.profile(False)
            print("[bench stream prof]", {k: round(v[0] / max(1, v[1]), 4) for k, v in eng.profile_get().items()}, file=sys.stderr)
            eng.profile_reset(); eng.profile(True)
            timed_region(step_resident(batches), 0, 12)
            eng.profile(False)
            print("[bench resident prof]", {k: round(v[0] / max(1, v[1]), 4) for k, v in eng.profile_get().items()}, file=sys.stderr)
            eng.profile_reset()
        extras["streaming"] = dict(region_line(el, n, k_s), ratio_to_resident=round((n / el) / value, 4),
                                   feed_build=a.feed_build, feed_ahead=a.feed_ahead,
                                   what="every step's batch arrives from page-locked host memory: id validation + identical-prefix plan + occurrence "
                                        "index derived per batch (host worker threads, or kernels on a side stream with --feed-build device) and "
                                        "uploaded under the steps in flight")
    if world == 1 and not a.force_dp and a.stream_total_paths > 0 and not main_streaming:
        # a path set of the size the config names, streamed: ceil(total / paths per step) steps, every batch fed from host memory under the steps in flight
        k_t = int(np.ceil(a.stream_total_paths / float(np.mean(paths_of))))
        for i in range(12):
            step_streaming(i)
        eng.sync()
        el, n = timed_region(step_streaming, 12, k_t)
        extras["streaming_total"] = dict(region_line(el, n, k_t), total_paths=int(n), seconds=round(el, 3), ratio_to_resident=round((n / el) / value, 4),
                                         feed_build=a.feed_build, feed_ahead=a.feed_ahead,
                                         what="the whole path set streamed once: every step's batch arrives from page-locked host memory (validation, plan, "
                                              "occurrence index derived per batch on host worker threads, one upload per batch under the steps in flight)")
    if plain and not main_streaming:
        # enough steps for a >= 0.3 s timed region
        k_long = int(min(4000, max(a.steps, np.ceil(0.35 / max(elapsed / a.steps, 1e-6)))))
        # ... with HIP events around EVERY fused family (the headline region keeps only the dominant family's: an event pair costs ~4 us of stream
        # time; three pairs in a >= 0.3 s region of ~1.4 ms steps are < 1 %): roofline.other_timed_families
        long_fams_on = prof and dominant.startswith("lstm_fused")
        if long_fams_on:
            eng.profile_reset(); eng.set_option("profile_filter", "lstm_fused"); eng.profile(True)
        el, n = timed_region(step, a.warmup + a.steps, k_long)
        if long_fams_on:
            eng.profile(False)
            extras["long_fams"] = (eng.profile_get(), a.warmup + a.steps, k_long)
            eng.profile_reset(); eng.set_option("profile_filter", dominant)
        extras["long_run"] = region_line(el, n, k_long)
        # every step of every path executed (no identical-prefix plan)
        eng.set_option("prefix_plan", "0")
        full = [eng.batch(hi, hl) for hi, hl in host]
        eng.set_option("prefix_plan", "1")
        st = step_resident(full)
        for i in range(3):
            st(i)
        eng.sync()
        el, n = timed_region(st, 3, a.steps)
        extras["no_prefix_plan"] = region_line(el, n, a.steps)
        for b in full:
            b.free()
    if plain and not main_streaming and not a.no_batch_sweep and not a.score_only and not a.train_only:
        # the reference's own minibatch regime (run_scripts/config.sh:38: 128 pairs ~ 256 paths; test_from_checkpoint.lua:49: 512) up to the
        # bench's 65 536: same step (scoring forward + trainBatch), batches resident, bucketed by P like the main region
        sweep = {}
        for pps in (256, 1024, 4096, 16384, 65536):
            pool = []
            for i, P in enumerate(Ps):
                pairs = max(1, pps // P)
                idx, labels = synth.make_paths(pairs, P, T, F=F, Vt=Vt, Ve=Ve, Vr=Vr, num_types=nT, seed=4242 + 13 * i)
                pool.append(eng.batch(idx, labels))
            npaths = [b.n_paths for b in pool]
            def st(i, pool=pool, npaths=npaths):
                run_batch(pool[i % len(pool)])
                return npaths[i % len(pool)]
            for i in range(6):
                st(i)
            eng.sync()
            k = int(min(2000, max(12, 0.12 / max(1e-6, (elapsed / a.steps) * max(pps / a.paths_per_step, 0.02)))))
            k -= k % len(pool)
            el, n = timed_region(st, 6, k)
            sweep[str(pps)] = {"value": round(n / el, 1), "ms_per_step": round(1e3 * el / k, 4), "steps": k}
            for b in pool:
                b.free()
        for v in sweep.values():
            v["ratio_to_65536"] = round(v["value"] / sweep["65536"]["value"], 4)
        extras["batch_sweep"] = {"unit": "paths/s", "paths_per_step": sweep,
                                 "what": "scoring forward + trainBatch per step at smaller batches (256 ~ config.sh:38's 128 pairs); resident batches, bucketed by P"}
    if plain and not main_streaming and not a.no_batch_sweep and not a.score_only and not a.train_only:
        extras["dropin_minibatch"] = dropin_minibatch(eng, opt, T, F, Vt, Ve, Vr, nT)
    # data-parallel runs: the exchange's own time, from a short extra region with events around its parts
    dp_info = None
    if dpx is not None:
        dpx.timing = True
        eng.profile_reset()
        eng.set_option("profile_filter", "dp_")
        eng.profile(True)
        k_x = min(a.steps, 12)
        timed_region(step, a.warmup + a.steps, k_x)
        eng.profile(False)
        fx = eng.profile_get()
        dp_info = {"world": world, "capacity_rows": dpx.capacity, "packed_MB_per_rank": round((4 + dpx.capacity * (1 + de_)) * 4 / 1e6, 2),
                   "exchange_ms_per_step": dpx.timing_summary(),
                   "pack_rows_ms": round(fx.get("dp_pack_rows", (0, 1))[0] / max(1, fx.get("dp_pack_rows", (0, 1))[1]), 4),
                   "merge_rows_ms": round(fx.get("dp_merge_rows", (0, 1))[0] / max(1, fx.get("dp_merge_rows", (0, 1))[1]), 4)}
        dpx.timing = False
        tr = torch.ones(1, device=dev)
        if world > 1:
            dist.all_reduce(tr)
        dp_info["ranks_reporting"] = int(tr.item())
        # self-check of a scaling run: the communicator's size as RCCL reports it, and every rank's own wall time per step of the timed region
        dp_info["rccl_nranks"] = int(eng.dp_comm_size()) if dpx.native else int(dist.get_world_size())   # ncclCommCount of the engine's own communicator
        mine = torch.tensor([1e3 * elapsed_local / a.steps], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        dp_info["per_rank_ms_per_step"] = [round(float(x.item()), 4) for x in allr]
        dp_info["fallback_reason"] = dpx.fallback_reason   # why the engine's own exchange is not in use (None: it is)
        # ---- the north_star's scaling experiment beside the weak line: 1 M paths strong-scaled, and the same total on rank 0 alone
        if not a.total_paths and not a.no_strong_1m and a.dims == "A" and a.compute_dtype == 0 and not a.score_only and not a.train_only:
            try:
                dp_info["strong_1M"] = strong_1m(a, eng, dpx, opt, run_batch, timed_region, barrier, world, rank, local_rank, stream,
                                                 dict(T=T, F=F, Vt=Vt, Ve=Ve, Vr=Vr, nT=nT, dt=dt_, de=de_, dr=dr_, H=H, L=L, C=C), Ps)
            except dp.DpHang:
                raise
            except Exception as ex:   # noqa: BLE001  (deterministic failures happen on every rank alike: the weak line is still worth printing)
                dp_info["strong_1M"] = {"error": f"{type(ex).__name__}: {ex}"}
        # Replicas must hold the same bits after the timed steps: the rows are summed in rank order inside the row update and the dense arena in rank
        # order inside the merge, so parameters AND Adam state are bit-identical by construction.  A 64-bit digest of the flat
        # parameter vector and of both Adam moments from every rank; any difference fails the run (exit code 3 below).
        eng.sync()
        per_rank, same = replica_digests((eng.get_flat_params(), eng.get_flat_opt_state(0), eng.get_flat_opt_state(1)), world, dev)
        dp_info["replica_digests"] = {"what": "blake2b-64 of (flat parameters, Adam m, Adam v) per rank", "per_rank": per_rank}
        dp_info["replicas_bit_identical"] = same
        dp_info["exchange"] = ("engine: in-place RCCL all-gather on the engine's stream, union inside the row update" if dpx.native else
                               "torch.distributed collectives around the pack / merge hooks" + ("" if a.dp_unfused else ", union inside the row update"))
        dp_info["scoring_pass"] = ("first, beside the training forward" if dp_score_first else
                                   (f"split: {1 - dp_split:.2f} of its tiles first (beside the training forward), {dp_split:.2f} behind the backward, under the all-gather"
                                    if 0.0 < dp_split < 1.0 else "behind the backward, under the all-gather"))

    loss = eng.read_loss()
    assert np.isfinite(loss), "training diverged"
    if os.environ.get("KPRN_BENCH_HOST_TIMING") and host_t["feed"]:
        f, r = np.array(host_t["feed"][-a.steps:]) * 1e3, np.array(host_t["run"][-a.steps:]) * 1e3
        print(f"[bench host timing] feed call ms mean {f.mean():.3f} max {f.max():.3f} | step queueing ms mean {r.mean():.3f} max {r.max():.3f}", file=sys.stderr)
        print("[bench host timing] feed:", np.round(f[:24], 2).tolist(), file=sys.stderr)
        print("[bench host timing] run: ", np.round(r[:24], 2).tolist(), file=sys.stderr)

    # ---- roofline of the dominant kernel family (HIP events recorded inside the timed region)
    roofline = None
    kernels = {}
    if fams:
        # algorithmic work per family over the timed steps
        for name, (ms, launches) in fams_warm.items():   # all families: from the warm-up steps
            kernels[name] = {"ms": round(ms, 4), "launches": launches, "from": "warmup"}
        for name, (ms, launches) in fams.items():        # the dominant family: live, inside the timed region
            kernels[name] = {"ms": round(ms, 4), "launches": launches, "from": "timed"}
        def family_roofline(name, ms, launches, first=None, k=None):
            # launches of a family all see the same N within a step; average work per launch over the region's steps (first, k: the headline region)
            first = a.warmup if first is None else first
            k = a.steps if k is None else k
            work = 0.0
            for i in range(k):
                N = paths_of[(first + i) % len(batches)]
                fw = family_work(name, N, T, D, H, L, C, F, nT, dt_, de_, dr_, G, exec_of[(first + i) % len(batches)])
                if fw is None:
                    return None
                work += fw[1]
            if launches <= 0 or ms <= 0:
                return None
            bound = family_work(name, 1, T, D, H, L, C, F, nT, dt_, de_, dr_, G)[0]
            # every launch of a family inside one step sees that step's N; launches per step is constant
            total_work = work * (launches / k)
            if bound == "mfma":
                achieved = total_work / (ms * 1e-3) / 1e12
                peak, unit = (PEAK_TFLOPS_BF16_MFMA if (a.compute_dtype == 1 and "bf16" in name) else PEAK_TFLOPS_F32_MFMA), "TFLOP/s"
            else:
                achieved = total_work / (ms * 1e-3) / 1e9
                peak, unit = PEAK_HBM_GBS, "GB/s"
            r = {"kernel": name, "bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                 "frac": round(achieved / peak, 4), "traffic": None, "avg_launch_ms": round(ms / launches, 5),
                 "launches": launches, "algorithmic_work_per_launch": round(total_work / launches)}
            tr = pmc_traffic(name, a.paths_per_step)
            if tr:
                r["traffic"] = tr["hbm_bytes_per_launch"]
                r["traffic_source"] = tr["source"]
            return r
        name, (ms, launches) = max(fams.items(), key=lambda kv: kv[1][0])
        roofline = family_roofline(name, ms, launches)
        if roofline is None:
            roofline = {"kernel": name, "bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None,
                        "traffic": None, "avg_launch_ms": round(ms / max(launches, 1), 5), "launches": launches}
        if len(fams) > 1:   # the other families timed inside the region (the persistent bf16 layer kernel's scoring launch beside its training launch)
            roofline["other_timed_families"] = {n: family_roofline(n, m, l) for n, (m, l) in fams.items() if n != name}
        elif "long_fams" in extras:
            # the fused forward launches (and the dominant family once more), timed with their own events inside the long_run region.  The scoring
            # forward runs on the side stream beside the training forward: its event pair spans its wait for CUs as well (see kernels_note).
            lf, lfirst, lk = extras["long_fams"]
            roofline["other_timed_families"] = {n: dict(family_roofline(n, m, l, lfirst, lk) or {}, region="long_run", steps=lk) for n, (m, l) in sorted(lf.items())}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a, T, dt_, de_, dr_, H, L, a.cpu_seconds)

    # ---- not the headline: the same workload with the forward on the bf16 matrix cores (compute_dtype 2, "f32x6": every fp32
    #      operand split exactly into three bf16 pieces, six partial products, fp32 accumulation; error <= an fp32 FMA chain;
    #      held to the same parity tolerances by tests/test_gpu_parity.py).  A short second run on a second engine.
    alt = None
    if rank == 0 and world == 1 and a.compute_dtype == 0 and not a.no_alt and not shipped and a.dims == "A" and not a.force_dp \
            and not a.score_only and not a.train_only and a.impl == "auto":
        eng2 = _ffi.Engine(Vt, Ve, Vr, dt_, de_, dr_, H, L, F=F, num_types=nT, C_=C, reducer=2, device_id=local_rank, rank=rank, world=world,
                           param_init=0.1, seed=12345, stream=stream, compute_dtype=2)
        eng2.set_option("score_overlap", "0" if a.no_score_overlap else "1")
        b2 = []
        for i, P in enumerate(Ps):
            pairs = max(1, a.paths_per_step // P)
            idx, labels = synth.make_paths(pairs, P, T, F=F, Vt=Vt, Ve=Ve, Vr=Vr, num_types=nT, seed=12345 + 97 * i + 7919 * rank)
            b2.append(eng2.batch(idx, labels))
        def step2(i):
            b = b2[i % len(b2)]
            eng2.forward_async(b, 1)
            eng2.train_step(b, opt, 1, want_loss=False)
            return b.n_paths
        for i in range(a.warmup):
            step2(i)
        eng2.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n2 = 0
        k2 = a.steps
        for i in range(k2):
            n2 += step2(a.warmup + i)
        eng2.sync()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t0
        alt = {"value": round(n2 / e2, 1), "unit": "paths/s", "ms_per_step": round(1e3 * e2 / k2, 4), "steps": k2, "dtype": "f32x6",
               "final_loss": round(float(eng2.read_loss()), 6),
               "what": "same workload, forward GEMMs on the bf16 matrix cores from an exact 3-way bf16 split of every fp32 operand "
                       "(6 partial products, fp32 accumulate; backward unchanged); opt-in via kprn_config.compute_dtype = 2"}
        for b in b2:
            b.free()
        eng2.close() if hasattr(eng2, "close") else None

    # ---- the other BASELINE configurations, compact: each a short run of this script in its own process (its own engine, its own
    #      tables), after everything above has been measured.  Parity-test configs, not the headline.
    other = None
    if rank == 0 and world == 1 and a.dims == "A" and a.compute_dtype == 0 and not a.no_other_configs and not a.force_dp and not a.score_only \
            and not a.train_only and a.impl == "auto" and not a.no_extra_regions:
        other = other_configs()

    if rank == 0:
        fwd_flops = sum(T * 2 * G * H * ((D if l == 0 else H) + H) for l in range(L)) + 2 * H * C
        step_flops = (0 if a.score_only else 3 * fwd_flops) + (0 if a.train_only else fwd_flops)  # nominal: every path, every step
        exec_frac = sum(exec_of) / float(sum(p * T for p in paths_of))
        # flops the timed steps really executed (fused-kernel accounting of family_work: skipped identical steps and the
        # absent recurrent half of a path's first executed step are NOT counted)
        exec_flops = 0.0
        for i in range(a.steps):
            bi = (a.warmup + i) % len(batches)
            fw = family_work("lstm_fused_fwd", paths_of[bi], T, D, H, L, C, F, nT, dt_, de_, dr_, G, exec_of[bi])[1]
            bw = family_work("lstm_fused_bwd", paths_of[bi], T, D, H, L, C, F, nT, dt_, de_, dr_, G, exec_of[bi])[1] * L
            exec_flops += (0 if a.score_only else fw + bw) + (0 if a.train_only else fw)
        exec_tflops = exec_flops * world / elapsed / 1e12
        # Products the small-table identity removes from layer 0's backward: the type / relation thirds of dx and of dW_i2g are
        # replaced by ns one-hot columns of the merged dW product + two tiny fp32 products.  `mfma_frac_end_to_end` above prices the model's ALGORITHMIC
        # flops (what the reference computes); `..._executed` only the MFMA work the engine really issues.
        saved = 0.0
        gh = G * H
        generic_tabs = (shipped or a.dims == "B" or a.impl == "generic" or (c4 and a.compute_dtype != 1)) and not a.score_only and not gru
        st_on = not any(o.replace(" ", "") in ("small_tables=0", "bf16_small_tables=0") for o in a.set_option)
        if st_on and not a.score_only and (c4 and a.compute_dtype == 1 or generic_tabs) and nT == 1:
            ns = 128 if (c4 and a.compute_dtype == 1) else ((Vr + Vt + 3) // 4) * 4
            if ns <= (128 if (c4 and a.compute_dtype == 1) else dt_):
                for i in range(a.steps):
                    Np = paths_of[(a.warmup + i) % len(batches)]
                    saved += 2.0 * T * Np * gh * (D - de_) + 2.0 * T * Np * gh * (D - (ns + de_))
        exec_tflops_issued = (exec_flops - saved) * world / elapsed / 1e12
        wl = (f"C2 KKBOX-MI synthetic: T={T}, D=H={H} ({dt_}/{de_}/{dr_}), L={L} FastLSTM, fp32, Ve={Ve}, "
              f"C=46, LSE pool, Adam; scoring pass + train step per batch")
        if gru:
            wl = (f"config.sh's sizes with -rnnType gru (OneModel.lua:237-238), synthetic KKBox-shaped paths: T={T}, nn.GRU, D={D} ({dt_}/{de_}/{dr_}), H={H}, L=1, "
                  f"fp32, Ve={Ve}, C=46, LSE pool, Adam; scoring pass + train step per batch")
        if shipped:
            wl = (f"run_scripts/config.sh as shipped, synthetic KKBox-shaped paths: T={T}, rnn (ReLU, MaskZero, identity init), "
                  f"D={D} ({dt_}/{de_}/{dr_}), H={H}, L=1, fp32, Ve={Ve}, C=46, LSE pool, Adam; scoring pass + train step per batch")
        if c4:
            wl = (f"C4 (BASELINE configs[3]) synthetic KG: {Ve} entities / {Vr} relations, T={T}, d=128 -> D=H={H}, L=1 FastLSTM, "
                  + ("bf16 storage (shadow tables / weights, activations, gate saves) + bf16 MFMA, fp32 accumulate / cell state / master parameters / Adam"
                     if a.compute_dtype == 1 else "fp32") + ", C=46, LSE pool; scoring pass + train step per batch")
        out = {
            "metric": ("paths/sec (train+score) at path_len=6 d=64" if not c4 else "paths/sec (train+score) at path_len=6 d=128 bf16"),
            "value": round(value, 1), "unit": "paths/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4),
            "higher_is_better": True, "scaling": "strong" if a.total_paths else "weak", "vs_baseline": None,
            "dtype": {0: "f32", 1: "bf16", 2: "f32x6", 3: "f32x3"}[a.compute_dtype], "data": "synthetic",
            "config": {"workload": wl,
                       "paths_per_step_per_gpu": a.paths_per_step, "paths_per_pair_buckets": Ps, "impl": a.impl,
                       "entity_update": "lazy-exact" if a.entity_update == 0 else "dense",
                       "score_overlap": not a.no_score_overlap,
                       "batch_feed": "streaming" if main_streaming else "resident",
                       "total_paths": a.total_paths or None,
                       "requested_steps": requested_steps, "paths_per_rank_step": a.paths_per_step,   # (--total-paths lowers the step count: strong_steps())
                       "uniform_tiles": bool(a.uniform_tiles), "set_options": a.set_option or None,
                       "forward_arithmetic": {0: "fp32 MFMA", 1: "bf16 MFMA products, fp32 accumulate",
                                              2: "f32x6: fp32 operands split exactly into 3 bf16 pieces, 6 partial products on the matrix cores, fp32 accumulate",
                                              3: "f32x3: pre-scaled fp32 operands as 2 fp16 pieces, 3 partial products on the matrix cores, fp32 accumulate"}[a.compute_dtype],
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "executed_step_fraction": round(exec_frac, 4),  # (path, step) positions executed / nominal: identical leading (pad) steps run once per batch
            "model_tflops_nominal": round(value * step_flops / 1e12, 3),   # as if every step of every path were computed
            "executed_tflops": round(exec_tflops, 3),
            # executed flops / wall clock / the MFMA peak of the type the products are formed in
            "mfma_frac_end_to_end": round(exec_tflops / ((PEAK_TFLOPS_BF16_MFMA if (c4 and a.compute_dtype == 1) else PEAK_TFLOPS_F32_MFMA) * world), 4),
            "mfma_frac_end_to_end_executed": round(exec_tflops_issued / ((PEAK_TFLOPS_BF16_MFMA if (c4 and a.compute_dtype == 1) else PEAK_TFLOPS_F32_MFMA) * world), 4),
            "small_table_identity": ({"on": True, "flops_not_issued_frac": round(saved / max(exec_flops, 1.0), 4),
                                      "what": "layer 0's type / relation gradients from G = dA^T [S_r | S_t] (one-hot columns of the merged dW product): the type / "
                                              "relation thirds of dx and dW_i2g are not computed; `mfma_frac_end_to_end` counts the model's algorithmic flops, "
                                              "`mfma_frac_end_to_end_executed` the issued ones"} if saved > 0 else None),
            "final_loss": round(loss, 6),
            "roofline": roofline, "cpu_baseline": cpu,
            "streaming": extras.get("streaming"), "streaming_total": extras.get("streaming_total"), "long_run": extras.get("long_run"), "batch_sweep": extras.get("batch_sweep"),
            "dropin_minibatch": extras.get("dropin_minibatch"),
            "other_configs": other,
            "value_no_prefix_plan": (extras.get("no_prefix_plan") or {}).get("value"), "no_prefix_plan": extras.get("no_prefix_plan"),
            "dp": dp_info, "alt_f32x6": alt, "kernels": kernels,
            "kernels_note": ("ms = HIP events around the family's launches, summed over `launches`; 'timed' = inside the timed region, 'warmup' = in the warm-up steps "
                             "(first steps of the process: lower clocks).  A family queued on a side stream beside a persistent launch that owns every CU "
                             "(loss_stage / head_bwd / bf16_transposes / entity_grad) reports ELAPSED time including its wait for CUs, not its own run time: "
                             "the families do not add up to ms_per_step"),
        }
    # a data-parallel run whose replicas diverged is not a measurement: the line is still printed (it says which rank differs), the exit code is 3
    diverged = bool(dp_info is not None and not dp_info.get("replicas_bit_identical", True))
    if world > 1 or a.force_dp:
        dist.destroy_process_group()
    if rank == 0:
        emit_last(out, hard_exit=(world > 1 or a.force_dp), rc=3 if diverged else 0)
    elif diverged:
        sys.exit(3)


STRONG_TOTAL = 1_000_000   # north_star: ">= 70 % data-parallel scaling efficiency at 8 GPUs on 1M synthetic length-6 paths"


def strong_1m(a, eng, dpx, opt, run_batch, timed_region, barrier, world, rank, local_rank, stream, shp, Ps):
    """The strong-scaling experiment inside a weak-scaling invocation (the driver passes no --total-paths): STRONG_TOTAL paths divided over the
    ranks and over strong_steps() steps through the same data-parallel step as the headline, and the SAME total on rank 0 alone (a second engine,
    the plain step, the other ranks waiting at a barrier) -- so that speed-up and efficiency come from one invocation on one set of boxes."""
    import torch
    from kprn_amd import _ffi, synth
    T, F, Vt, Ve, Vr, nT = shp["T"], shp["F"], shp["Vt"], shp["Ve"], shp["Vr"], shp["nT"]

    def make_pool(engine, pps, seed0):
        pool = []
        for i, P in enumerate(Ps):
            idx, labels = synth.make_paths(max(1, pps // P), P, T, F=F, Vt=Vt, Ve=Ve, Vr=Vr, num_types=nT, seed=seed0 + 97 * i)
            pool.append(engine.batch(idx, labels))
        return pool

    # (a) all ranks: total / (world * steps) paths per rank and step
    ks = strong_steps(STRONG_TOTAL, world, a.steps)
    pps = max(64, STRONG_TOTAL // (world * ks))
    pool = make_pool(eng, pps, 555 + 7919 * rank)
    npaths = [b.n_paths for b in pool]
    cap_before = dpx.capacity
    dpx.set_capacity(max(max(b.n_uniq for b in pool), cap_before))   # (collective: the larger of the two bounds, the same on every rank)

    def st(i):
        run_batch(pool[i % len(pool)])
        return npaths[i % len(pool)]
    for i in range(2):
        st(i)
    el, n = timed_region(st, 2, ks)
    out = {"total_paths": STRONG_TOTAL, "steps": ks, "paths_per_rank_step": pps, "value": round(n / el, 1), "unit": "paths/s",
           "ms_per_step": round(1e3 * el / ks, 4), "paths_counted": int(n)}
    for b in pool:
        b.free()
    # (b) rank 0 alone: the same total through the plain step of a second engine (no exchange), enough steps for >= 32 768 paths each
    k1 = strong_steps(STRONG_TOTAL, 1, a.steps)
    pps1 = max(64, STRONG_TOTAL // k1)
    n1 = None
    barrier()
    if rank == 0:
        e1 = _ffi.Engine(Vt, Ve, Vr, shp["dt"], shp["de"], shp["dr"], shp["H"], shp["L"], F=F, num_types=nT, C_=shp["C"], reducer=2, device_id=local_rank,
                         rank=0, world=1, param_init=0.1, seed=12345, stream=stream)
        e1.set_option("score_overlap", "0" if a.no_score_overlap else "1")
        p1 = make_pool(e1, pps1, 555)
        np1 = [b.n_paths for b in p1]

        def st1(i):
            b = p1[i % len(p1)]
            e1.forward_async(b, 1)
            e1.train_step(b, opt, 1, want_loss=False)
            return np1[i % len(p1)]
        for i in range(3):
            st1(i)
        e1.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        c1 = sum(st1(3 + i) for i in range(k1))
        e1.sync(); torch.cuda.synchronize()
        e1s = time.perf_counter() - t0
        n1 = {"steps": k1, "paths_per_step": pps1, "value": round(c1 / e1s, 1), "ms_per_step": round(1e3 * e1s / k1, 4), "paths_counted": int(c1),
              "what": "rank 0 alone: a second engine, the plain step (no exchange), the other ranks waiting"}
        for b in p1:
            b.free()
        e1.close()
    barrier()
    if n1 is not None:
        out["n1"] = n1
        out["speedup_vs_n1"] = round(out["value"] / n1["value"], 4)
        out["efficiency"] = round(out["value"] / n1["value"] / world, 4)
    out["what"] = ("strong scaling measured inside this invocation: value = total paths / max-over-ranks time of `steps` data-parallel steps; n1 = the same "
                   "total on rank 0 alone; efficiency = value / (n1.value * n_gpus).  Steps are lowered until a rank's step holds >= 32 768 paths")
    return out


def emit_last(out, hard_exit=False, rc=0):
    """The JSON line is the LAST thing on stdout.  RCCL prints a version banner through C stdio, which a pipe buffers until the process
    exits -- after Python's own buffer, i.e. behind the line: flush C stdio first, print, and (runs that initialised RCCL only) leave without
    running exit handlers.  Other runs exit normally: rocprofv3 writes its tables from an exit handler."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
    sys.stderr.flush()
    if hard_exit and os.environ.get("KPRN_BENCH_SOFT_EXIT") != "1":   # (a profiler wants its exit handler: scripts/gpu_timeline.sh)
        os._exit(rc)
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
