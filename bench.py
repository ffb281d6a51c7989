#!/usr/bin/env python3
"""Benchmark of the FDTD field-update path on MI355X.

One "step" is one full time step (H, D, source, E, Mur) of one slab.  At
--gpus 1 the workload is the headline configuration: a 16384 x 16384 2-D TMz
grid, vacuum, one point source at the centre.  At --gpus N every rank holds a
16384 x 16384 slab of a 16384 x (16384*N) grid (weak scaling) and exchanges halo
rows with each neighbour over RCCL on a second stream, beside the sweep of the
interior rows (once per pass of the multi-step sweep: eight rows of every array;
once per step, one ez row, with the one-step kernels).  The metric is
Mcell-updates/s = cells * steps / seconds / 1e6, whole job.

Prints one JSON line (rank 0):
  roofline           the dominant kernel (HBM bound), timed with HIP events on the library's
                     stream: `achieved` = algorithmic bytes of ONE launch / its average duration.
                     A pass of the multi-step sweep moves every live array once whatever the
                     number of steps it advances: 49 B per cell (25 in, 24 out); only launches
                     that advanced the full steps_per_pass are averaged, `launches` lists all.
                     `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes
                     (profiles/traffic.json); `one_step_equivalent` prices the same cell-updates
                     at the 41 B a one-step sweep moves; `issue_bound` is a vector-issue estimate.
  roofline_one_step  (one GPU, 2-D) the one-step sweep on the same workload: 41 B per cell-update.
  cpu_baseline       (one GPU) the CPU oracle on the host cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_CELL_2D = 41   # read hx hy ez^n ez^n-1 dz^n dz^n-1 (24) + code (1); write hx hy ez dz (16)
BYTES_PER_CELL_PASS_2D = 49  # multi-step sweep, per cell and PASS: the same reads (25); write hx hy + two levels of ez, dz (24)
BYTES_PER_CELL_3D = 97   # read 3 H + 6 E + 6 D (60) + code (1); write 3 H + 3 E + 3 D (36)
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak


def cpu_baseline(nx, ny, nz=1, budget_s=20.0):
    """Oracle (CPU restatement, OpenMP over rows) on a bounded sample: the same
    vacuum + centre-source workload on a smaller grid for about budget_s seconds."""
    from oracle import oracle
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    p = oracle.make_params(nx, ny, nz, [(nx // 2, ny // 2, nz // 2)])
    o = oracle.Oracle(p, ecoef=oracle.ECOEF_CONSISTENT)
    o.step(3 if nz > 1 else 10, threads)                 # touch the pages, spin up the thread pool
    probe = 5 if nz > 1 else 20
    t_probe = o.time_steps(probe, threads) / probe
    steps = max(3, min(5000, int(budget_s / max(t_probe, 1e-6))))
    sec = o.time_steps(steps, threads)
    shape = f"{nx}x{ny}x{nz} 3-D" if nz > 1 else f"{nx}x{ny} 2-D TMz"
    return {
        "value": nx * ny * nz * steps / sec / 1e6,
        "unit": "Mcell-updates/s",
        "cores": threads,
        "kind": "port",
        "sample": f"CPU oracle (C, -O2 -ffp-contract=off, OpenMP), {shape} vacuum + centre source, "
                  f"{steps} steps, {sec:.1f} s",
    }


KERNEL_SOURCES = {"fdtd2d_sweep_tb": "sweep2d_tb.hip", "fdtd2d_sweep": "sweep2d.hip", "fdtd2d_tile": "sweep2d_tile.hip",
                  "fdtd3d_sweep": "sweep3d.hip", "fdtd3d_sweep_tb": "sweep3d_tb.hip"}


def kernel_source_sha16(kernel):
    """Fingerprint of the source a kernel is built from (its .hip file and the two headers every
    kernel includes): a committed PMC measurement is only quoted for the code it was taken with."""
    import hashlib
    h = hashlib.sha256()
    for name in (KERNEL_SOURCES.get(kernel, ""), "device_math.h", "internal.h"):
        try:
            with open(os.path.join(ROOT, "fdtd_amd", "csrc", name), "rb") as f:
                h.update(f.read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def measured_traffic(workload, kernel, spp):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes, if any -- and only if the
    kernel's source is still the one the passes were taken with (None otherwise: a stale figure is
    worse than none)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(f"{workload} [{kernel} x{spp}]", {})
    except (OSError, ValueError):
        return None
    if entry.get("kernel_source_sha16") != kernel_source_sha16(kernel):
        return None
    return entry.get("hbm_bytes_per_launch")


def issue_bound(solver, k, kernel_ms):
    """Vector-issue estimate of one pass of the multi-step sweep: vector instructions of a row
    iteration (counted in the gfx950 assembly by tools/isa_count.py: fp32/integer 2 cycles, fp64 4
    cycles per wave64 instruction) x row iterations of the launch / (1024 SIMDs x 2.4 GHz)."""
    try:
        with open(os.path.join(ROOT, "profiles", "isa_counts.json")) as f:
            c = json.load(f)[f"K{k}_one_plain"]["inner_march_per_row_iteration"]
    except (OSError, ValueError, KeyError):
        return None
    waves, iters = solver.sweep_work
    ms = iters * c["valu_issue_cycles"] / (1024 * 2.4e9) * 1e3
    return {"valu_f32_per_row_iteration": c["valu_f32"], "valu_f64_per_row_iteration": c["valu_f64"],
            "valu_issue_cycles_per_row_iteration": c["valu_issue_cycles"], "waves": waves, "row_iterations": iters,
            "model_ms": ms, "frac_of_kernel_time": ms / kernel_ms}


def one_step_leg(params, device, cells):
    """The one-step sweep (one pass over memory per time step: 41 algorithmic bytes per cell-update)
    on the same workload, timed the same way: 5 warm-up steps, 20 timed."""
    import numpy as np
    import fdtd_amd
    from fdtd_amd import binding as B
    s = fdtd_amd.Solver(params, ecoef=B.ECOEF_CONSISTENT, kernel=B.KERNEL_FUSED, device=device)
    s.step(5)
    s.profile_begin(20)
    s.step(20)
    ms, _ = s.profile_read_launches(20)
    avg = float(np.mean(ms))
    achieved = cells * BYTES_PER_CELL_2D / (avg * 1e-3) / 1e9
    name = s.sweep_kernel_name
    out = {"kernel": name, "kernel_avg_ms": avg, "launches_averaged": int(len(ms)), "algorithmic_bytes_per_cell_update": BYTES_PER_CELL_2D,
           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "Mcell_updates_per_s": cells / (avg * 1e-3) / 1e6,
           "traffic": measured_traffic(f"{params.nx}x{params.ny} 2-D TMz vacuum + centre point source", name, 1)}
    s.close()
    return out


class stdout_to_stderr:
    """Libraries (gloo, RCCL) print banners on fd 1 while they initialise; stdout is reserved
    for the one JSON line, so fd 1 points at stderr inside this block."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nx", type=int, default=16384)
    ap.add_argument("--ny", type=int, default=16384, help="rows per GPU")
    ap.add_argument("--nz", type=int, default=1, help="> 1 selects the 3-D path (z planes per GPU)")
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--kernel", choices=["auto", "split", "fused", "multistep", "tile"], default="auto")
    ap.add_argument("--steps-per-pass", type=int, default=0, help="multi-step kernels: time steps per pass (2..8; 0 = the library's choice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device", type=int, default=-1, help="GPU ordinal (default: LOCAL_RANK)")
    ap.add_argument("--snapshot-every", type=int, default=0,
                    help="take an ez snapshot (side-stream D2H) every N steps inside the timed region")
    ap.add_argument("--probes", type=int, default=1, help="probes recorded every step (1..16), spread over the grid")
    ap.add_argument("--allow-host-halo", action="store_true",
                    help="N > 1: if RCCL cannot initialise, run with halos staged through the host (flagged in the output) "
                         "instead of failing")
    ap.add_argument("--no-profile", action="store_true",
                    help="no HIP-event pairs around the sweeps (two more host calls per launch: they dominate the step "
                         "time of grids that are launch-bound); the roofline block is then left out")
    ap.add_argument("--no-one-step-leg", action="store_true", help="skip the extra one-step-sweep measurement")
    ap.add_argument("--tissue", type=int, default=0, metavar="U",
                    help="config 3: the 265x490 body slice data/pgmdata/v1_00800.pgm upsampled U times (32: 8480x15680), "
                         "mat_specs_riken tissues, Mur with the default rp, source in the air beside the body (1 GPU)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --ny (or --nz) is the extent of the WHOLE grid, cut into --gpus slabs")
    args = ap.parse_args()

    import numpy as np
    import torch

    import fdtd_amd
    from fdtd_amd import binding as B

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.device >= 0:
        local_rank = args.device
    torch.cuda.set_device(local_rank)

    dist = None
    if world > 1:
        import torch.distributed as dist
        # control plane (rendezvous, barrier, max over ranks) on gloo; the data plane
        # is the library's own RCCL communicator (ncclSend/ncclRecv of halo rows)
        with stdout_to_stderr():
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    is3d = args.nz > 1
    if args.tissue:
        if world > 1 or is3d:
            raise SystemExit("--tissue is a one-GPU 2-D workload")
        args.nx, args.ny = 265 * args.tissue, 490 * args.tissue
    nx = args.nx
    if args.strong:
        if (args.nz if is3d else args.ny) % world:
            raise SystemExit("--strong needs the slab axis to be a multiple of --gpus")
        if is3d:
            args.nz //= world
        else:
            args.ny //= world
    if is3d:
        gdims = (nx, args.ny, args.nz * world)
        slab = (rank * args.nz, args.nz)
        src = [(nx // 2, args.ny // 2, gdims[2] // 2)]
        cells_per_rank = nx * args.ny * args.nz
        bytes_per_cell = BYTES_PER_CELL_3D
    else:
        gdims = (nx, args.ny * world, 1)
        slab = (rank * args.ny, args.ny)
        src = [(40 * args.tissue, 245 * args.tissue, 0)] if args.tissue else [(nx // 2, gdims[1] // 2, 0)]
        cells_per_rank = nx * args.ny
        bytes_per_cell = BYTES_PER_CELL_2D
    params = fdtd_amd.make_params(*gdims, src)
    kernel = {"auto": B.KERNEL_AUTO, "split": B.KERNEL_SPLIT, "fused": B.KERNEL_FUSED,
              "multistep": B.KERNEL_MULTISTEP, "tile": B.KERNEL_TILE}[args.kernel]
    if world > 1 and kernel == B.KERNEL_AUTO:
        # every rank must run the same kernel with the same halo depth: say so instead of leaving
        # it to each rank's own decision
        kernel = B.KERNEL_FUSED if is3d else B.KERNEL_MULTISTEP
        if not is3d and args.steps_per_pass == 0:
            args.steps_per_pass = 8
    solver = fdtd_amd.Solver(params, ecoef=B.ECOEF_CONSISTENT, kernel=kernel, device=local_rank,
                             slab=slab if world > 1 else None, chunk_rows=args.chunk_rows,
                             steps_per_pass=args.steps_per_pass)
    if args.tissue:
        data = os.path.join(ROOT, "data")
        solver.load_materials(os.path.join(data, "mat_specs_riken"), os.path.join(data, "pgmdata"), first_slice=800,
                              upsample=args.tissue)
    spp = solver.steps_per_pass          # time steps one full launch of the sweep advances
    halo = "none"
    halo_fallback = False
    stepper = solver
    if world > 1:
        # Halo exchange inside the library over RCCL.  Should RCCL not come up on this node the run
        # fails -- unless --allow-host-halo, with which every rank falls back to the host-staged
        # exchange (same kernels, halo rows moved by gloo) and the JSON line says so at top level.
        err = ""
        try:
            ids = [fdtd_amd.binding.comm_unique_id() if rank == 0 else None]
        except fdtd_amd.FdtdError as e:
            ids, err = [None], str(e)
        dist.broadcast_object_list(ids, src=0)
        if ids[0] is not None:
            try:
                with stdout_to_stderr():
                    solver.comm_init(ids[0], rank, world)
                    solver.comm_selftest()
            except fdtd_amd.FdtdError as e:
                err = str(e)
        flags = [None] * world
        dist.all_gather_object(flags, err)
        if any(flags):
            msg = f"bench.py: RCCL unavailable ({[f for f in flags if f][0]})"
            if not args.allow_host_halo:
                raise SystemExit(msg + "; pass --allow-host-halo to measure with the host-staged exchange instead")
            if rank == 0:
                print(msg + "; host-staged halo exchange", file=sys.stderr)
            stepper = fdtd_amd.SlabRunner(solver, rank, world, transport="host")
            halo_fallback = True
            halo = "host-staged over gloo (RCCL failed to initialise), 1 E row per neighbour per step"
        else:
            halo = "RCCL send/recv on a second stream beside the sweep of the interior rows, 1 E row per neighbour per step"
        if solver.steps_per_pass > 1:
            halo = halo.replace("1 E row per neighbour per step",
                                f"{solver.steps_per_pass} rows of 6 arrays + codes per neighbour per pass")

    src_rank = min(world - 1, src[0][2 if is3d else 1] // (args.nz if is3d else args.ny))
    # probe 0 sits on the source (it feeds the sanity check below), the others lie on the diagonal
    nprobes = max(1, min(16, args.probes))
    probes = [src[0]] + [tuple(int(d * (i + 0.5) / nprobes) if d > 1 else 0 for d in gdims) for i in range(1, nprobes)]
    solver.set_probes(probes, capacity=args.steps + args.warmup)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stepper.step(args.warmup)
    if args.snapshot_every > 0:
        # both staging sets (device copy + pinned host buffer) are allocated by the first two snapshots
        for _ in range(2):
            solver.snapshot_async(B.EZ)
        solver.snapshot_wait()
    if not args.no_profile:
        solver.profile_begin(args.steps)
    barrier()
    t0 = time.perf_counter()
    snapshots = 0
    if args.snapshot_every > 0:
        done = 0
        while done < args.steps:
            n = min(args.snapshot_every, args.steps - done)
            stepper.step(n)
            done += n
            if done % args.snapshot_every == 0:
                solver.snapshot_async(B.EZ)            # waits for the previous one, then queues the next
                snapshots += 1
        solver.sync()                                  # the steps are done: what is left is the last D2H
        t_tail = time.perf_counter()
        solver.snapshot_wait()
        tail_ms = (time.perf_counter() - t_tail) * 1e3
    else:
        tail_ms = 0.0
        stepper.step(args.steps)
    solver.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    sweep_ms, sweep_steps = solver.profile_read_launches(args.steps) if not args.no_profile else ([], [])

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    # the run must have moved real data: the field next to the source is finite and non-zero
    # (read through the probe ring: 16 values, not a copy of the whole field)
    series = solver.read_probes(args.warmup, args.steps)
    if rank == src_rank:
        ez_src = series[:, 0, -1]
        assert np.all(np.isfinite(ez_src)) and np.any(ez_src != 0), f"ez at the source is zero or not finite: {ez_src[-3:]}"

    if rank == 0:
        total_cells = cells_per_rank * world
        value = total_cells * args.steps / elapsed / 1e6
        kernel_name = solver.sweep_kernel_name
        workload = (f"{gdims[0]}x{gdims[1]}x{gdims[2]} 3-D" if is3d else f"{gdims[0]}x{gdims[1]} 2-D TMz") + \
            (f" tissue map (v1_00800.pgm x{args.tissue}, mat_specs_riken) + Mur + point source" if args.tissue
             else " vacuum + centre point source")
        # Launches of the main sweep inside the timed region, grouped by the time steps they advanced:
        # fdtd_step(n) runs full passes and one shorter remainder pass (a single step: the one-step sweep).
        groups = {}
        for ms, k in zip(sweep_ms, sweep_steps):
            groups.setdefault(int(k), []).append(float(ms))
        launches = [{"steps": k, "count": len(v), "avg_ms": float(np.mean(v))} for k, v in sorted(groups.items(), reverse=True)]
        full = groups.get(spp) or next(iter(groups.values()), [float("nan")])
        full_k = spp if spp in groups else (launches[0]["steps"] if launches else spp)
        sweep_avg_ms = float(np.mean(full))
        multistep = spp > 1 and not is3d
        # Algorithmic bytes of ONE launch: every live array once in and once out.  A pass of the
        # multi-step sweep reads levels n, n-1, H and the codes (25 B per cell) and writes levels
        # n+K, n+K-1 and H (24 B) whatever K is; a one-step sweep moves 41 B (2-D) / 97 B (3-D).
        bytes_per_cell_launch = BYTES_PER_CELL_PASS_2D if multistep else bytes_per_cell
        achieved = cells_per_rank * bytes_per_cell_launch / (sweep_avg_ms * 1e-3) / 1e9
        traffic = measured_traffic(workload, kernel_name, full_k)
        out = {
            "metric": "Mcell-updates/s (E+H)",
            "value": value,
            "unit": "Mcell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "halo_fallback": halo_fallback,
            "config": {
                "workload": workload,
                "cells_per_gpu": cells_per_rank,
                "ecoef": "consistent",
                "kernel": kernel_name,
                "steps_per_pass": spp,
                "parallelism": f"slab{world}" if world > 1 else "single",
                "halo": halo,
                "output": (f"{snapshots} ez snapshots per rank (async D2H on a side stream) + {nprobes} probes x "
                           f"{solver.lib.fdtd_probe_fields(solver._ctx)} fields every step; the timed region ends with "
                           f"{tail_ms:.1f} ms of waiting for the last snapshot's D2H") if snapshots
                else f"{nprobes} probe(s) every step",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": kernel_name,
                "kernel_avg_ms": sweep_avg_ms,
                "steps_per_launch": full_k,
                "launches_averaged": len(full),
                "algorithmic_bytes_per_cell_per_launch": bytes_per_cell_launch,
                "algorithmic_bytes_per_launch": cells_per_rank * bytes_per_cell_launch,
                "launches": launches,
            },
        }
        if args.no_profile:
            del out["roofline"]
            multistep = False
        if traffic and not args.no_profile:
            out["roofline"]["traffic_GBs"] = traffic / (sweep_avg_ms * 1e-3) / 1e9
            out["roofline"]["traffic_frac_of_peak"] = traffic / (sweep_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if multistep:
            # what the same cell-updates would cost a sweep that advances one step per pass
            eq = cells_per_rank * full_k * BYTES_PER_CELL_2D / (sweep_avg_ms * 1e-3) / 1e9
            out["roofline"]["one_step_equivalent"] = {
                "bytes_per_cell_update": BYTES_PER_CELL_2D, "GB/s": eq, "ratio_to_peak": eq / HBM_PEAK_GBS,
                "note": f"{full_k} steps per launch: 41 B x cell-updates / time; exceeds the peak because a pass "
                        "moves 49 B per cell whatever the number of steps"}
            issue = issue_bound(solver, full_k, sweep_avg_ms) if kernel_name == "fdtd2d_sweep_tb" else None
            if issue:
                out["roofline"]["issue_bound"] = issue
        if world == 1 and not is3d and not args.no_one_step_leg and not args.no_profile and not args.tissue \
                and kernel_name != "fdtd2d_sweep":
            solver.close()
            out["roofline_one_step"] = one_step_leg(params, local_rank, cells_per_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(4096, 4096) if not is3d else cpu_baseline(256, 256, 256)
        print(json.dumps(out), flush=True)

    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
