#!/usr/bin/env python3
"""bench.py -- physics steps/sec at fixed dt = 1/60 s, 100k bodies (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one PhysicsWorld::think(1/60) (/root/reference/gui_client/PhysicsWorld.cpp:1356-1443) over the synthetic
BASELINE config 3 world: 100k mixed box / sphere / capsule bodies (100x100x10 lattice, seed 3, substrata_amd/scenes.py)
dropped on the ground quad.  All state is resident in HBM before the timed region; every step blocks until the device
has finished it, like think().  N > 1: weak scaling, one 100k-body tile per GPU side by side (4x2 for 8), one fused RCCL
all-gather of ghost bodies per step (substrata_amd/tiles.py); value = N x world-steps/s = 100k-body tile-steps per second.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the world's stream
(sgp_world_step_profiled); `cpu_baseline` times the CPU oracle (a port, NOT Jolt) on a bounded sample of the same
workload: the device state after warm-up is copied into the oracle and a few steps are timed on one host core.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DT = 1.0 / 60.0
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SWEEP_BYTES_PER_BODY = 188     # integrate + AABB body-array sweep, 112 B read + 76 B written
# per contact point per velocity iteration: 12 precomputed row vectors (3 axes x 4 float4) + lambdas read, lambdas written
SOLVE_BYTES_PER_POINT = 12 * 16 + 16 + 16
SOLVE_BYTES_PER_MANIFOLD = 28 + 2 * 32 + 2 * 32  # ab 8 + normal/friction 16 + np 4; the velocity halves of two solver-body records read and written
# HBM traffic of the three sweep kernels per step at 100k bodies from the rocprofv3 PMC passes in profiles/r01m_pmc_hbm_traffic.md
# (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE): 35.4 MB read + 26.0 MB written per step
SWEEP_TRAFFIC_BYTES_PER_BODY_PMC = 614.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=120)
    ap.add_argument("--bodies", type=int, default=100000, help="bodies per tile (BASELINE: 100k)")
    ap.add_argument("--workload", default="config3", choices=["config3", "config5"],
                    help="config3 = the bench line (100k mixed bodies); config5 = 1k cars + 50k debris (extra measurement, N=1 only)")
    ap.add_argument("--cpu-steps", type=int, default=16, help="oracle steps timed for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the oracle (0 = min(32, host cpus))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=8)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for single-GPU dry runs of the tile exchange)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    from substrata_amd import abi, scenes, tiles
    from substrata_amd.lib import World, init

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    dist = None
    if args.share_gpu:
        local_rank = 0
    if world_size > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
        assert world_size == n_gpus, "launch with --nproc-per-node equal to --gpus"
    elif n_gpus != 1:
        raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run with N ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    init()

    # ---- world: config 3 tile(s) -------------------------------------------------------------------------------
    nx = ny = 100
    nz = max(1, args.bodies // (nx * ny))
    spacing = 1.5
    tile_w, tile_d = nx * spacing, ny * spacing
    lo, hi, origin = tiles.tile_bounds(rank, n_gpus, tile_w, tile_d)
    descs = scenes.config3_100k_mixed(nx, ny, nz, seed=3 + rank)
    # tile-local lattice is centred on the origin: move it to the tile's place
    descs["pos"][1:, 0] += origin[0] + tile_w / 2 - spacing / 2
    descs["pos"][1:, 1] += origin[1] + tile_d / 2 - spacing / 2
    car_ids = []
    if args.workload == "config5":
        if n_gpus != 1:
            raise SystemExit("--workload config5 is a single-GPU measurement")
        descs, car_ids = scenes.config5_cars_debris()
    n_bodies = len(descs) - 1
    w = World(max_bodies=len(descs) + 32768, device=local_rank)
    if len(car_ids):
        scenes.use_car_hull(w, descs, car_ids)       # chassis = the reference's 12-point convex hull with a lowered centre of mass
    w.add_batch(descs)
    for b in car_ids:
        w.vehicle_create(w.default_vehicle_desc(int(b)))
    sim_step = [0]
    xdev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")
    ex = tiles.GhostExchange(w, rank, n_gpus, lo, hi, margin=2.0, dist=dist, device=xdev) if n_gpus > 1 else None

    def one_step():
        if ex is not None:
            ex.exchange()
        if len(car_ids):                 # driver input arrives every frame (CarPhysics::update -> SetDriverInput)
            w.vehicle_set_inputs(0, scenes.config5_inputs(len(car_ids), sim_step[0] * DT))
        sim_step[0] += 1
        w.step(DT)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    contacts = 0
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = w.stats()

    # ---- the application's real loop: every step followed by the read-back of the active bodies' poses ------
    # (GUIClient walks activated_obs after think(), GUIClient.cpp:6581-6723).  Reported next to the headline, never as `value`.
    n_rb = min(args.steps, 60)
    barrier()
    rb_buf = np.empty(n_bodies + 8, dtype=abi.body_state_dtype)
    t1 = time.perf_counter()
    for _ in range(n_rb):
        one_step()
        active_states = w.read_active(out=rb_buf)
    barrier()
    readback_steps_per_s = n_rb / (time.perf_counter() - t1) if n_rb else 0.0
    n_read_back = len(active_states) if n_rb else 0

    # ---- roofline: HIP events around every launch, same world, steps right after the timed region -----------------
    names = w.kernel_class_names()
    ksum = np.zeros(len(names)); klaunch = np.zeros(len(names)); n_prof = max(1, args.profile_steps)
    pts = cons = 0
    for _ in range(n_prof):
        if ex is not None:
            ex.exchange()
        p = w.step_profiled(DT)
        ksum += np.array([p.kernel_ms[k] for k in range(len(names))])
        klaunch += np.array([p.kernel_launches[k] for k in range(len(names))])
        pts += p.num_contact_points; cons += p.num_constraints
        sweep_bodies = p.sweep_bodies
    k = {nm: i for i, nm in enumerate(names)}
    sweep_ms = (ksum[k["apply_forces"]] + ksum[k["integrate_pose"]] + ksum[k["finalize"]]) / n_prof
    sweep_bytes = SWEEP_BYTES_PER_BODY * sweep_bodies
    sweep_gbs = sweep_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    sv = k["solve_velocity"]
    solve_launch_ms = ksum[sv] / max(klaunch[sv], 1)
    iters = w.desc.settings.num_velocity_steps
    solve_bytes_per_launch = (SOLVE_BYTES_PER_POINT * pts / n_prof + SOLVE_BYTES_PER_MANIFOLD * cons / n_prof) * iters / max(klaunch[sv] / n_prof, 1)
    solve_gbs = solve_bytes_per_launch / (solve_launch_ms * 1e-3) / 1e9 if solve_launch_ms > 0 else 0.0

    out = None
    if rank == 0:
        steps_per_s = args.steps / elapsed
        out = {
            "metric": "physics steps/sec at fixed dt, 100k bodies" if args.workload == "config3" else "physics steps/sec at fixed dt, 1k cars + 50k debris",
            "value": steps_per_s * n_gpus,
            "unit": "steps/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE config 3: 100k mixed box/sphere/capsule bodies, 100x100x10 lattice spacing 1.5 m, seed 3, "
                             "ground quad 2000 m, dt 1/60, Jolt default settings (10 velocity / 2 position iterations), sleeping enabled")
                            if args.workload == "config3" else
                            ("BASELINE config 5: 1024 cars (32x32 grid, spacing 8 m; chassis = the 12-point convex hull of the car script, centre of mass lowered 0.2 m, 1200 kg, 4 wheels, "
                             "FWD, Scripting.cpp defaults; input forward=1, steer=sin(0.5t+id) refreshed every step) + 50k unit-box debris, seed 5, dt 1/60"),
                "bodies_per_gpu": n_bodies, "tiles": n_gpus, "value_definition": "n_gpus x world steps/s (one 100k-body tile per GPU)",
                "active_bodies_end": st.num_active, "contact_constraints_end": st.num_manifolds,
                "contact_points_end": st.num_contact_points, "colours_end": st.num_colours,
                "ghosts_exported_per_step": (ex.last_exported if ex else 0), "ghosts_imported_per_step": (ex.last_imported if ex else 0),
                "dropped_pairs_or_manifolds": st.pairs_dropped + st.manifolds_dropped,
                "steps_per_s_with_active_pose_readback": readback_steps_per_s * n_gpus, "bodies_read_back_per_step": n_read_back,
            },
            "roofline": {
                "bound": "hbm", "kernel": "body-array sweep = k_apply_forces + k_integrate_pose + k_finalize (one launch each per step)",
                "achieved": sweep_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sweep_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": sweep_bytes, "launch_ms": sweep_ms,
                "traffic": SWEEP_TRAFFIC_BYTES_PER_BODY_PMC * sweep_bodies,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/r01m_pmc_hbm_traffic.md (tools/collect_pmc.sh); float4-padded SoA, the pose read by all three passes, sleep-test spheres and the pose write-back move 3.3x the algorithmic bytes",
            },
            "roofline_solver": {
                "bound": "hbm", "kernel": "k_solve_velocity (dominant by time; one launch per colour per iteration)",
                "achieved": solve_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": solve_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": solve_bytes_per_launch, "launch_ms": solve_launch_ms,
                "launches_per_step": klaunch[sv] / n_prof, "traffic": None,
            },
            "kernel_ms_per_step": {names[i]: round(ksum[i] / n_prof, 4) for i in range(len(names)) if klaunch[i]},
        }

    # ---- CPU baseline: the oracle (a port of the same step, NOT Jolt) on a bounded sample, rank 0 only --------------
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and args.cpu_steps > 0:
        from oracle import oracle
        S = w.read_states(0, len(descs))
        d2 = descs.copy()
        d2["pos"] = S["pos"]; d2["rot"] = S["rot"]; d2["lin_vel"] = S["lin_vel"]; d2["ang_vel"] = S["ang_vel"]
        d2["activate"] = (S["active"] != 0).astype(np.int32)
        threads = args.cpu_threads or min(32, os.cpu_count() or 1)
        threads = oracle.set_threads(threads)
        cw = oracle.OracleWorld(max_bodies=len(descs) + 8)
        if len(car_ids):
            cw.hull_create(scenes.CAR_HULL_POINTS, com_offset=scenes.CAR_COM_OFFSET)     # same hull id as on the device
        cw.add_batch(d2)
        for b in car_ids:                # (drivetrain state starts fresh on the CPU side: same cost per step, not the same trajectory)
            cw.vehicle_create(cw.default_vehicle_desc(int(b)))
        if len(car_ids):
            cw.vehicle_set_inputs(0, scenes.config5_inputs(len(car_ids), sim_step[0] * DT))
        cw.step(DT)                      # builds the contact cache so the timed steps are warm-started like the device's
        t1 = time.perf_counter()
        for _ in range(args.cpu_steps):
            cw.step(DT)
        cpu_el = time.perf_counter() - t1
        cst = cw.stats()
        oracle.set_threads(1)
        t2 = time.perf_counter()
        for _ in range(2):
            cw.step(DT)
        cpu1 = 2 / (time.perf_counter() - t2)
        out["cpu_baseline"] = {
            "value": args.cpu_steps / cpu_el, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": f"{args.cpu_steps} steps of the same {n_bodies}-body world, started from the device state after warm-up + timed region "
                      f"({cst.num_manifolds} contact constraints, {cst.num_active} active bodies); oracle/sgo_oracle.c with {threads} OpenMP "
                      f"threads (single thread: {cpu1:.2f} steps/s); this is this repo's CPU restatement, not JoltPhysics (absent from the reference tree)",
            "host_cpus": os.cpu_count(),
        }
        cw.close()
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    w.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
