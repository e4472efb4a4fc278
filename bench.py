#!/usr/bin/env python3
"""bench.py — CLIPPER dense-cluster hot path on MI355X: affinity build + solve().

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic registration problem:
  affinity build (scorePairwiseConsistency, EuclideanDistance) + solve() (findDenseClique)
on BASELINE.json's headline configuration: m = 10 000 putative associations, 95 % outliers,
synthetic 3-D points (SURVEY.md 8d recipe, clipper_amd/synth.py). The inputs (D1, D2, A,
u0) are staged into HBM before the timed region; every step re-runs the affinity kernel and
the full solver. With N > 1 the SAME problem is column-sharded over the N GPUs (one process
per GPU, per-pass RCCL all-gather of the (M_off x, C_off x) slices): strong scaling.

`value` is the step with the inputs resident in HBM when the timed region starts — the measurement
contract of this build ("if the boundary hands over host buffers, note the PCIe-inclusive rate ...
it is never `value`"); the same step with host buffers handed to the drop-in entry points (H2D of
D1, D2, A, u0 and D2H of u inside: SURVEY 8d's wording of the metric) is timed over the same number
of warmed steps and reported beside it as `ms_per_step_host_buffers`.

Prints ONE JSON line on rank 0 (see the contract in the task statement), extended with
  "roofline"     achieved GB/s of the dominant kernel (the pass: one sweep over M for the candidates of the
                 line-search window in use - DESIGN.md 3a: candidate 0 alone while line searches accept their
                 first trial, which covers every pass on M of the headline; all six otherwise), from HIP
                 events recorded on the solver stream around every 20th launch in the timed region (an event
                 pair costs ~30 us of stream time).
                 Default storage "csc" (one GPU): k_gemv_slices streams the slices of M — bytes per
                 launch = what the slices hold (headers, lengths, value and row quads), NOT
                 s*m^2; `useful_bytes_per_launch` / `frac_useful` count stored entries only
                 (value + row byte, no quad padding); the dense-equivalent rate is not a
                 roofline figure. `regime` says whether those bytes fit the 256 MiB Infinity
                 Cache (then every pass after the first is served on-die and "hbm" names the
                 peak it is normalised by, not the wire it crossed). `traffic` = HBM-side bytes
                 per launch from the PMC run recorded in profiles/pmc_r05.json
                 (`traffic_source` names the entry and the commit it was measured at).
                 --storage f32: k_gemv on the dense store, bytes per launch = 4*m*W
  "roofline_affinity"  the fill kernel: bytes of M written per build / its duration.
  "roofline_resident"  k_solve_view_resident, the launch that runs the iterations on the row view inside the chip's LDS
                 (25 of the headline's 36 passes): its duration (device wall clock in every timed step; HIP events
                 around it in one extra, untimed solve), its iterations, the LDS bytes an iteration reads against
                 the LDS peak, and what its waves do meanwhile (PMC record: LDS-busy and waiting shares).
  "cpu_baseline" the oracle (oracle/libclipper_ref.so, a port of the reference) timed on
                 this box's host cores on the same problem, median of 3 (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PMC_RECORD = "profiles/pmc_r06.json"
LDS_PEAK_TBPS = 150.0   # 256 CUs x 256 B/clk x ~2.4 GHz for ds_read_b64 / b128 (MI355X_MICROARCH.md, LDS)
# the sources of the kernels the PMC record is about (the pass, its decision, the planner that cuts its work, the
# fill): a record is quoted only if it was measured on exactly these bytes
PMC_SOURCES = ("clipper_amd/csrc/k_slices.hip.h", "clipper_amd/csrc/k_solver.hip.h", "clipper_amd/csrc/host_plan.hpp",
               "clipper_amd/csrc/k_affinity.hip.h", "clipper_amd/csrc/k_csc.hip.h", "clipper_amd/csrc/k_rv_resident.hip.h",
               "clipper_amd/csrc/k_resident.hip.h", "clipper_amd/csrc/k_rowview.hip.h", "clipper_amd/csrc/k_gemv.hip.h",
               "clipper_amd/csrc/k_subproblem.hip.h", "clipper_amd/csrc/k_matrix.hip.h", "clipper_amd/csrc/host_rv_resident.hpp",
               "clipper_amd/csrc/host_resident.hpp", "clipper_amd/csrc/host_rowview.hpp", "clipper_amd/csrc/host_subproblem.hpp")


def kernel_sources_sha256():
    import hashlib
    h = hashlib.sha256()
    for rel in PMC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--m", "--size", dest="m", type=int, default=10000,
                    help="putative associations (under torch.distributed.run write --size: it reads --m as one of its own options)")
    ap.add_argument("--rho", type=float, default=None, help="outlier ratio (default 0.95; 0.90 at m<=1000)")
    ap.add_argument("--storage", choices=["f32", "f64", "csc", "csc64"], default="csc",
                    help="how M is kept in HBM: csc / csc64 = stored entries only (fp32 / fp64 values; the "
                         "solver's passes skip the zeros), f32 / f64 = dense (vectors/accumulators are always f64)")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true",
                    help="do not bracket mat-vec launches with HIP events (roofline fields become 0)")
    ap.add_argument("--probe-m", type=int, default=100000,
                    help="size of the `scaling_probe` object: the same step at the size whose pass is "
                         "HBM-bound and whose shards scale (0 = no probe)")
    ap.add_argument("--probe-steps", type=int, default=3)
    ap.add_argument("--rank-path", action="store_true",
                    help="(testing) with one GPU: the multi-process construction of --gpus N > 1 in a 1-rank world — "
                         "clipper_hip_create_rank, ncclCommInitRank, every exchange through ncclAllGather — so that the "
                         "code path the driver's multi-GPU runs take can be exercised on a one-GPU box")
    ap.add_argument("--sources-sha", action="store_true", help="print the sha256 of the kernel sources a PMC record is tied to, and exit")
    return ap.parse_args()


def cpu_baseline(problem, args):
    """Oracle (port of the reference) on the host cores: OpenMP affinity loop on all cores,
    single-threaded sparse self-adjoint solver — the reference's own threading model."""
    from clipper_amd import synth
    from oracle import clipper_ref as ref

    reps = 3 if args.m <= 12000 else 1
    runs = []
    for _ in range(reps):
        r = ref.RefClipper()
        t0 = time.perf_counter()
        r.score_pairwise_consistency_euclidean(problem.D1, problem.D2, problem.A,
                                               **synth.EUCLID_BENCH_PARAMS)
        t1 = time.perf_counter()
        s = r.solve(problem.u0)
        t2 = time.perf_counter()
        runs.append((t2 - t0, t1 - t0, t2 - t1))
    runs.sort()
    tot, ta, ts = runs[len(runs) // 2]   # the median step
    return {
        "value": round(tot * 1e3, 3), "unit": "ms", "cores": ref.omp_threads(),
        "kind": "port",
        "sample": (f"median of {reps} full steps at m={args.m}: affinity {1e3 * ta:.1f} ms on "
                   f"{ref.omp_threads()} OpenMP threads + solve {1e3 * ts:.1f} ms on 1 thread "
                   f"({s.n_passes} reference-counted passes, nnz={r.nnz}); all steps "
                   f"{[round(1e3 * x[0], 1) for x in runs]} ms"),
        "affinity_ms": round(ta * 1e3, 3), "solve_ms": round(ts * 1e3, 3),
        "nproc": os.cpu_count(),
    }, s


def main():
    args = parse()
    if args.sources_sha:
        print(kernel_sources_sha256())
        return
    N = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != N:
        if world == 1 and N > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        N = world
    rho = args.rho if args.rho is not None else (0.90 if args.m <= 1000 else 0.95)

    # torch first: it must bind its own bundled ROCm runtime (importing it after another
    # libamdhip64 is already loaded leaves torch.cuda without devices — measured on the GPU
    # box); the product library then shares that runtime. torch is only plumbing here:
    # barrier / synchronize / max-over-ranks and the broadcast of the RCCL unique id.
    import torch
    import torch.distributed as dist

    torch.cuda.init()
    from clipper_amd import _abi as abi
    from clipper_amd import synth

    abi.load_library()

    from clipper_amd import dist as cdist

    if N > 1:
        torch.cuda.set_device(local_rank)
        cdist.init_process_group("cpu:gloo,cuda:nccl")

    def barrier_sync():
        if N > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    storage = {"f32": abi.STORE_F32, "f64": abi.STORE_F64, "csc": abi.STORE_F32_CSC,
               "csc64": abi.STORE_F64_CSC}[args.storage]
    problem = synth.make_euclidean_problem(args.m, rho, seed=args.seed)  # identical on every rank
    if N == 1 and args.rank_path:
        os.environ["CLIPPER_HIP_FORCE_RCCL"] = "1"   # a 1-rank world still goes through ncclAllGather
        cdist.init_process_group("gloo")
    if N > 1 or args.rank_path:
        g = abi.HipClipper(device=local_rank, storage=storage, rank=rank, world=N)
        uid = cdist.broadcast_bytes(g.unique_id() if rank == 0 else None, 128, src=0)
        g.comm_init(uid)   # ncclCommInitRank inside libclipper_hip.so (RCCL over xGMI)
    else:
        g = abi.HipClipper(device=local_rank, storage=storage)

    inv = synth.EUCLID_BENCH_PARAMS

    def timed_steps(prob, steps, warmup, profile):
        """W untimed + K timed steps (affinity build + solve) on `prob`, inputs resident in HBM first."""
        g.stage_inputs(prob.D1, prob.D2, prob.A)
        g.affinity_euclidean_staged(**inv)
        g.stage_u0(prob.u0)
        for _ in range(warmup):
            g.affinity_euclidean_staged(**inv)
            g.solve_staged()
        g.set_profiling(profile)
        r = dict(aff_ms=[], solve_ms=[], gemv_us=0.0, gemv_n=0, view_us=0.0, view_n=0, xchg_us=0.0, xchg_n=0,
                 res_us=0.0, res_iters=0, res_launches=0, res_giveups=0, sub_us=0.0, sub_n=0)
        barrier_sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ta = time.perf_counter()
            g.affinity_euclidean_staged(**inv)
            tb = time.perf_counter()
            r["sol"] = g.solve_staged()
            tc = time.perf_counter()
            r["aff_ms"].append((tb - ta) * 1e3)
            r["solve_ms"].append((tc - tb) * 1e3)
            tm = g.timings()
            vs = g.view_stats()
            r["gemv_us"] += tm.gemv_avg_us * tm.gemv_launches
            r["gemv_n"] += tm.gemv_launches
            r["view_us"] += vs.view_pass_avg_us * vs.view_pass_samples
            r["view_n"] += vs.view_pass_samples
            r["sub_us"] += vs.sub_pass_avg_us * vs.sub_pass_samples
            r["sub_n"] += vs.sub_pass_samples
            r["xchg_us"] += tm.exchange_avg_us * tm.exchange_samples
            r["xchg_n"] += tm.exchange_samples
            r["res_us"] += vs.resident_us
            r["res_iters"] += vs.resident_iterations
            r["res_launches"] += vs.resident_launches
            r["res_giveups"] += vs.resident_giveups
        barrier_sync()
        elapsed = time.perf_counter() - t0
        g.set_profiling(False)
        if N > 1:
            elapsed = cdist.max_over_ranks(elapsed)
        r["ms_per_step"] = elapsed * 1e3 / steps
        r["tm"], r["vs"] = g.timings(), g.view_stats()
        return r

    R = timed_steps(problem, args.steps, args.warmup, not args.no_profile)
    sol, tm, vstats = R["sol"], R["tm"], R["vs"]
    aff_ms, solve_ms = R["aff_ms"], R["solve_ms"]
    ms_per_step = R["ms_per_step"]
    gemv_avg_us = R["gemv_us"] / max(1, R["gemv_n"])
    gemv_bytes = tm.gemv_bytes  # s * m * W_local: algorithmic bytes of ONE launch on THIS rank
    achieved = gemv_bytes / (gemv_avg_us * 1e-6) / 1e9 if gemv_avg_us > 0 else 0.0

    # PCIe-inclusive variant (host buffers handed to the drop-in entry points: H2D of the inputs
    # and D2H of u inside), same number of steps, warmed — for DESIGN.md, never `value`
    g.score_pairwise_consistency_euclidean(problem.D1, problem.D2, problem.A, **inv)
    g.solve(problem.u0)
    barrier_sync()
    th0 = time.perf_counter()
    for _ in range(args.steps):
        g.score_pairwise_consistency_euclidean(problem.D1, problem.D2, problem.A, **inv)
        sol_h = g.solve(problem.u0)
    barrier_sync()
    host_ms = (time.perf_counter() - th0) * 1e3 / args.steps
    if N > 1:
        host_ms = cdist.max_over_ranks(host_ms)

    # One more solve, untimed, with a HIP event pair around every launch of the resident solver on a view
    # (profiling level 2: the pair costs stream time, which is why it is not in the timed region)
    res_event = None
    if N == 1 and not args.no_profile:
        g.set_profiling(2)
        g.affinity_euclidean_staged(**inv)
        g.stage_u0(problem.u0)
        g.solve_staged()
        res_event = g.view_stats()
        g.set_profiling(False)

    # The same step at the size whose pass is HBM-bound and whose column shards scale:
    # recorded beside the headline for every N, so that a scaling run shows a curve that CAN scale
    def scaling_probe(pm, psteps):
        pp = synth.make_euclidean_problem(pm, 0.95, seed=args.seed)
        P = timed_steps(pp, psteps, 1, not args.no_profile)
        ptm, pvs = P["tm"], P["vs"]
        pass_us = P["gemv_us"] / max(1, P["gemv_n"])
        view_us = P["view_us"] / max(1, P["view_n"])
        sub_us = P["sub_us"] / max(1, P["sub_n"])

        def roof(nbytes, us):   # a streamed pass against the HBM peak: bytes its slices hold / its launch's duration
            if us <= 0 or nbytes <= 0:
                return None
            gbps = nbytes / (us * 1e-6) / 1e9
            return {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(gbps / HBM_PEAK_GBPS, 4), "bytes_per_launch": int(nbytes), "us_per_launch": round(us, 2)}
        return {
            "m": pm, "rho": 0.95, "steps": psteps, "n_gpus": N,
            "ms_per_step": round(P["ms_per_step"], 3),
            "affinity_ms": round(sum(P["aff_ms"]) / len(P["aff_ms"]), 3),
            "solve_ms": round(sum(P["solve_ms"]) / len(P["solve_ms"]), 3),
            "passes": int(P["sol"].n_passes), "trials": int(P["sol"].n_trials),
            "pass_on_M_us": round(pass_us, 1), "bytes_per_pass_on_M_this_rank": ptm.gemv_bytes,
            "pass_on_M_GBps": round(ptm.gemv_bytes / (pass_us * 1e-6) / 1e9, 1) if pass_us > 0 else 0.0,
            "pass_on_M_frac_of_hbm_peak": round(ptm.gemv_bytes / (pass_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if pass_us > 0 else 0.0,
            "passes_on_a_row_view": int(pvs.view_passes), "view_rows": int(pvs.rows), "view_bytes": int(pvs.bytes),
            "pass_on_view_us": round(view_us, 1),
            "roofline_view": roof(pvs.bytes, view_us),
            # the live sub-problem: the passes behind the view run on M[S,S], S = the associations
            # that can still be selected
            "passes_on_the_sub_problem": int(pvs.sub_passes), "sub_rows": int(pvs.sub_rows), "sub_bytes": int(pvs.sub_bytes),
            "sub_entries": int(pvs.sub_entries), "sub_leaves": int(pvs.sub_leaves), "sub_build_ms": round(pvs.sub_build_ms, 3),
            "sub_storage": "dense fp32 (k_gemv)" if pvs.sub_dense else "slices",
            "pass_on_sub_us": round(sub_us, 1),
            "roofline_sub": roof(pvs.sub_bytes, sub_us),
            "exchange_us": round(P["xchg_us"] / max(1, P["xchg_n"]), 1) if N > 1 else None,
            "exchange_bytes_per_rank": ptm.exchange_bytes if N > 1 else None,
            "nodes": int(len(P["sol"].nodes)), "score": P["sol"].score,
        }

    probe = None
    if args.probe_m and args.probe_m != args.m:
        probe = scaling_probe(args.probe_m, args.probe_steps)
    # BASELINE.json's configuration for the 8-GPU node (m = 300 000 row-sharded): recorded whenever there is more
    # than one rank (one GPU holds it too — 57 GB — but the default run must finish within minutes)
    probe_cfg5 = None
    if N >= 2 and args.probe_m and args.m != 300000:
        probe_cfg5 = scaling_probe(300000, 1)

    out = None
    if rank == 0:
        name, cus, hbm = g.device_info()
        in_use = {abi.STORE_F32: "f32", abi.STORE_F64: "f64", abi.STORE_F32_CSC: "csc",
                  abi.STORE_F64_CSC: "csc64"}[g.storage_in_use]
        compressed = in_use in ("csc", "csc64")
        # PMC figures cannot be collected inside this process (rocprofv3 wraps a command): they come
        # from profiles/pmc_r04.json, written by tools/pmc_summary.py from a rocprofv3 --pmc run of this
        # same command, with the commit, the kernel's byte count and the sha256 of the kernel sources at that
        # time. A record taken on other sources or other bytes than today's is NOT quoted.
        traffic, traffic_source, issue, resident_pmc = None, None, None, None
        pmc = os.path.join(ROOT, PMC_RECORD)
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(f"m{args.m}_{in_use}")
            except Exception:
                rec = None
            if rec:
                then = rec.get("pass_bytes_per_launch")
                sha_then, sha_now = rec.get("kernel_sources_sha256"), kernel_sources_sha256()
                stale_bytes = then is not None and abs(then - tm.gemv_bytes) > 1e-6 * max(1.0, tm.gemv_bytes)
                if stale_bytes or sha_then != sha_now:
                    print(f"bench.py: {PMC_RECORD} was measured on other kernel sources or bytes ({then} bytes per pass then, "
                          f"{tm.gemv_bytes} today; sources {str(sha_then)[:12]} then, {sha_now[:12]} today): STALE, not quoted — "
                          f"re-run tools/gpu_prof.sh", file=sys.stderr, flush=True)
                    traffic_source = {"file": PMC_RECORD, "stale": True, "kernel_bytes_then": then,
                                      "kernel_sources_sha256_then": sha_then, "kernel_sources_sha256_now": sha_now}
                else:
                    traffic = rec.get("pass_hbm_bytes_per_launch")
                    traffic_source = {"file": PMC_RECORD, "key": f"m{args.m}_{in_use}",
                                      "measured_at_commit": rec.get("commit"), "kernel_bytes_then": then,
                                      "kernel_sources_sha256": sha_then,
                                      "read_bytes": rec.get("pass_read_bytes"), "written_bytes": rec.get("pass_written_bytes"),
                                      "written_note": "what a pass writes is its partial sums, one slot per work item of a strip (28 at the "
                                                      "headline), written through (sc1) while the launch runs and read by k_tail: a pass on "
                                                      "candidate 0 alone writes 2 of the V + 1 rows of a slot (28 x 2 x 10 048 x 8 B = 4.5 MB "
                                                      "nominal, 4.0 MB counted), a pass on all six candidates 7 (15.8 MB nominal, 14.1 MB "
                                                      "counted in round 5 - not the < 5 MB the review estimated)",
                                      "how": rec.get("how")}
                    issue = rec.get("affinity_issue")
                    if issue and issue.get("SQ_ACTIVE_INST_VALU") and issue.get("simd_cycles_available"):
                        # the DIRECT busy counter (quad-cycles, summed over the SIMDs) against the launch's SIMD cycles: how much of
                        # the time a SIMD's VALU pipe is executing — the binding resource of the fill (NOTEBOOK.md, round 6)
                        issue["valu_pipe_active_frac"] = round(4.0 * issue["SQ_ACTIVE_INST_VALU"] / issue["simd_cycles_available"], 4)
                    resident_pmc = rec.get("resident")
        useful = tm.gemv_useful_bytes
        achieved_useful = useful / (gemv_avg_us * 1e-6) / 1e9 if gemv_avg_us > 0 else 0.0
        regime = ("Infinity-Cache-resident: %.0f MB per pass < 256 MiB, re-read every pass — served "
                  "on-die after the first pass; peak = HBM spec as the normalising figure" % (gemv_bytes / 1e6)
                  if gemv_bytes < 256 * 2 ** 20 else "HBM-streaming: %.2f GB per pass" % (gemv_bytes / 1e9))
        aff_kernel_ms = tm.affinity_kernel_ms
        aff_achieved = tm.affinity_bytes / (aff_kernel_ms * 1e-3) / 1e9 if aff_kernel_ms > 0 else 0.0
        roofline_resident = None
        if R["res_giveups"] > 0:  # (a give-up puts the context on the streamed route — the line would describe the fall-back)
            print(f"bench.py: {R['res_giveups']} launch(es) of the resident solver on a view GAVE UP during the timed steps: "
                  f"`value` was measured (partly) on the streamed fall-back", file=sys.stderr, flush=True)
        if R["res_launches"] > 0 and R["res_iters"] > 0:
            V = g.window
            ent = int(vstats.resident_entries)
            esz = 8 if in_use == "csc64" else 4
            # per stored entry and iteration: its value and row byte out of the unit's slices (quads of 4 values + 4 row
            # bytes) + ceil(V / 2) ds_read_b128 gathers of the candidates' rows of the X table (16 bytes each)
            lds_bytes = ent * (esz + 1.0 + 16.0 * ((V + 1) // 2))
            it_us = R["res_us"] / R["res_iters"]
            ach = lds_bytes / (it_us * 1e-6) / 1e12 if it_us > 0 else 0.0
            units = int(vstats.resident_units)
            peak_units = LDS_PEAK_TBPS * units / max(1, cus)
            roofline_resident = {
                "kernel": "k_solve_view_resident", "launches_per_solve": R["res_launches"] / args.steps,
                "gave_up": R["res_giveups"],
                "launch_us": round(R["res_us"] / R["res_launches"], 2),
                "launch_us_hip_events": (round(res_event.resident_event_us / max(1, res_event.resident_launches), 2)
                                         if res_event is not None and res_event.resident_launches > 0 else None),
                "iterations_per_launch": R["res_iters"] / R["res_launches"],
                "us_per_iteration": round(it_us, 3),
                "units": units, "view_rows": int(vstats.rows), "view_bytes": int(vstats.bytes), "view_entries": ent,
                "bound": "latency (a dependent chain per iteration: X table -> pass over the unit's slices in LDS -> "
                         "tail -> publish -> poll -> sums -> decision), not a pipe: see `waves`",
                "lds": {"bytes_per_iteration": lds_bytes, "achieved": round(ach, 2), "peak": round(peak_units, 1),
                        "unit": "TB/s", "frac": round(ach / peak_units, 4) if peak_units > 0 else None,
                        "peak_note": f"{LDS_PEAK_TBPS} TB/s chip-wide for ds_read_b64/b128 (256 B/clk/CU at 2.4 GHz) x {units}/{cus} CUs in use",
                        "bytes_note": "stored entries x (value + row byte + ceil(V/2) x 16-byte gathers of X rows)"},
                "waves": resident_pmc,   # PMC record: LDS-busy share of the launch's cycles, share of wave cycles spent waiting
                "how": "launch_us: the device's 100 MHz wall clock, unit 0's first instruction to the last unit's commit, "
                       "every timed step (free); launch_us_hip_events: hipEvent pair around the launch on its stream, one "
                       "extra untimed solve",
            }
        out = {
            "metric": "affinity+solve ms and GEMV HBM GB/s, n=10k assoc, 95% outliers, 1/8 GPU",
            "value": round(ms_per_step, 4),
            "unit": "ms",
            "n_gpus": N,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if in_use in ("f64", "csc64") else "f64 (M stored f32)",
            "data": "synthetic",
            "config": {
                "workload": (f"synthetic 3-D registration, m={args.m} putative associations, "
                             f"{int(round(rho * 100))}% outliers, EuclideanDistance"
                             f"{{sigma=0.015,epsilon=0.05}}, clipper::Params defaults, "
                             f"explicit u0 (seed {args.seed}+1)"),
                "m": args.m, "rho": rho, "storage": in_use,
                "parallelism": ("single GPU" if N == 1 and not args.rank_path else
                                f"column-sharded M over {N} GPU(s), one process each, RCCL all-gather per pass"),
                "device": name, "cus": cus,
            },
            "affinity_ms": round(sum(aff_ms) / len(aff_ms), 4),
            "solve_ms": round(sum(solve_ms) / len(solve_ms), 4),
            "affinity_kernel_ms": round(tm.affinity_kernel_ms, 4),
            "gemv_passes_per_solve": int(sol.n_passes),
            "line_search_trials_per_solve": int(sol.n_trials),
            "line_search_window": g.window,
            "gemv_avg_us": round(gemv_avg_us, 3),
            "gemv_min_us": round(tm.gemv_min_us, 3),
            "ms_per_step_host_buffers": round(host_ms, 4),
            "solution": {"score": sol.score, "nodes": int(len(sol.nodes)), "ifinal": int(sol.ifinal)},
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": "k_gemv_slices" if compressed else "k_gemv",
                "bytes_per_launch": gemv_bytes,
                "useful_bytes_per_launch": useful,
                "frac_useful": round(achieved_useful / HBM_PEAK_GBPS, 4),
                "regime": regime,
                "dense_equivalent_GBps": round(4.0 * args.m * args.m / (gemv_avg_us * 1e-6) / 1e9, 1)
                if (compressed and gemv_avg_us > 0) else None,
            },
            "roofline_affinity": {
                "bound": "valu-issue (VALU pipe active 0.59 of all SIMD cycles at m = 10k, 0.66 at 100k: `issue.valu_pipe_active_frac`)",
                "kernel": "k_affinity_sym (writes the slices itself)" if compressed else "affinity fill",
                "kernel_ms": round(aff_kernel_ms, 4),
                # VALU instructions the launch issued (SQ_INSTS_VALU) x 4 cycles per wave64 instruction on a
                # 16-lane fp64 / 32-lane fp32 SIMD-cycle budget = kernel time x SIMDs x clock: PMC record
                "issue": issue,
                "store_stream": {"bound": "hbm", "achieved": round(aff_achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": round(aff_achieved / HBM_PEAK_GBPS, 4), "bytes_per_launch": tm.affinity_bytes,
                                 "note": "what the fill writes, as a store floor only: it is bound by VALU issue "
                                         "(prefilter + exact fp64 scores of the survivors), see `issue`"},
            },
            "row_view": {
                "what": "passes on the slices of M[live rows, :] instead of M (rows with u = 0 and g <= 0 are exact "
                        "zeros in every line-search candidate; DESIGN.md 3c) - streamed by the pass kernel, or, for a "
                        "view of <= 1024 rows that fits the chip's LDS, run inside ONE resident launch (DESIGN.md 3d)",
                "passes_on_view": int(vstats.view_passes), "passes": int(sol.n_passes), "views_built": int(vstats.builds),
                "rows": int(vstats.rows), "bytes": int(vstats.bytes), "build_ms": round(vstats.build_ms, 4),
                "resident_launches": int(vstats.resident_launches), "resident_giveups": int(vstats.resident_giveups),
                # HIP events around streamed view passes; none when the view's iterations ran inside the resident launch
                "pass_on_view_us": round(R["view_us"] / R["view_n"], 2) if R["view_n"] > 0 else None,
            },
            # the pass on a streamed view against the HBM peak (none at the headline: its view's iterations run inside
            # the resident launch — `roofline_resident`; the probe carries the figure of the size that streams)
            "roofline_view": ({"bound": "hbm", "achieved": round(vstats.bytes / (R["view_us"] / R["view_n"] * 1e-6) / 1e9, 1),
                               "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(vstats.bytes / (R["view_us"] / R["view_n"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                               "bytes_per_launch": int(vstats.bytes), "us_per_launch": round(R["view_us"] / R["view_n"], 2)}
                              if R["view_n"] > 0 and vstats.bytes > 0 else None),
            "live_subproblem": {
                "what": "once a row view exists and the penalty d is large, no association outside a small set S can get a "
                        "positive gradient again (a bound on clipper.cpp:238-241 checked by every decision): the solve "
                        "continues on M[S,S] as a problem of its own (DESIGN.md 3e). m >= 12 000 only: below, the view's "
                        "iterations run inside the resident launch",
                "entries": int(vstats.sub_entries), "leaves": int(vstats.sub_leaves), "passes_on_it": int(vstats.sub_passes),
                "associations": int(vstats.sub_rows), "bytes": int(vstats.sub_bytes), "build_ms": round(vstats.sub_build_ms, 4),
                "pass_us": round(R["sub_us"] / R["sub_n"], 2) if R["sub_n"] > 0 else None,
            },
            "roofline_resident": roofline_resident,
            "scaling_probe": probe,
            "scaling_probe_cfg5": probe_cfg5,
        }
        if N == 1 and not args.no_cpu_baseline:
            cb, sref = cpu_baseline(problem, args)
            out["cpu_baseline"] = cb
            out["parity"] = {
                "set_identical": sref.nodes.tolist() == sol.nodes.tolist(),
                "rel_dscore": abs(sref.score - sol.score) / abs(sref.score),
            }
        print(json.dumps(out), flush=True)
    if N > 1 or args.rank_path:
        dist.destroy_process_group()
    g.close()


if __name__ == "__main__":
    main()
