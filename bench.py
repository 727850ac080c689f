"""bench.py -- fields-of-view/sec of the end-to-end segment + props hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W --batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = the whole config-3 chain (BASELINE.json configs[2]: Gaussian -> Otsu -> threshold -> open ->
close -> EDT -> peak markers -> watershed -> clear_border -> relabel -> morphology + 4-channel intensity
tables) over one batch of B synthetic 4 x 2048 x 2048 uint16 fields of view that are ALREADY RESIDENT in
HBM when the timed region starts.  For N > 1 every rank processes its own B fields of view (weak scaling)
and each step ends with the RCCL all-gather of the per-rank feature tables (BASELINE configs[3]).
Rank 0 prints ONE JSON line (contract in the task statement).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Each context owns up to three HIP streams (main + two fork/join streams for the flood classes) and the
# batch is split over several contexts; ROCm's default of 4 hardware queues per process would multiplex them
# (and with torch + RCCL in the process, 8 measured 14 % slower than 16).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (G/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
# algorithmic bytes per pixel per stage: narrowest dtypes, one read + one write
STAGE_BYTES_PER_PX = {
    "gaussian": 10, "otsu": 8, "threshold": 9, "opening": 4, "closing": 4, "threshold_open_close": 17, "label8": 5, "edt": 9, "peaks": 17,
    "markers": 5, "watershed": 17, "clear_border": 8, "relabel": 4, "regionprops": 16, "intensity": 12,
}


# kernels (name prefixes in the rocprofv3 output) that make up each stage of the chain
STAGE_KERNELS = {
    "gaussian": ("gauss_lds_kernel", "gauss_fused_kernel", "conv_v8_kernel", "conv_h8_kernel"),
    "otsu": ("hist_f64_kernel", "otsu_f64_kernel", "minmax_"),
    "threshold_open_close": ("pack_gt_kernel", "toc_fused_kernel", "packed_prim_kernel", "unpack_kernel"),
    "edt": ("edt_rows_kernel", "edt_cols_kernel"),
    "peaks": ("peaks_tile_kernel",),
    "markers": ("sp_",),
    "watershed": ("ccl_", "ws_", "roots_"),
    "label8": ("ccl_", "roots_", "apply_rank_kernel"),
    "clear_border": ("presence_", "frame_mark_kernel", "drop_flagged_kernel", "map_labels_kernel"),
    "regionprops": ("rp_",),
}


def pmc_traffic_bytes(stage: str):
    """HBM bytes per 32-FOV launch of one stage from the committed PMC summary (profiles/r01_hbm_pmc.csv:
    FETCH_SIZE with the gfx950 x2 correction for wide reads + WRITE_SIZE), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_hbm_pmc.csv")
    if not os.path.exists(path) or stage not in STAGE_KERNELS:
        return None
    total = 0.0
    with open(path) as f:
        for line in f:
            if line.startswith("#") or line.startswith("kernel,") or line.startswith("TOTAL"):
                continue
            parts = line.rstrip("\n").rsplit(",", 4)
            if len(parts) != 5:
                continue
            name = parts[0].strip('"').replace("void ", "")
            if any(name.startswith(pfx) for pfx in STAGE_KERNELS[stage]):
                total += (float(parts[3]) + float(parts[4])) * 1e6
    return total or None


def rocprof_kernel_us(stage: str):
    """Average duration (us per 32-FOV launch) of the kernels of one stage from the committed rocprofv3 summary
    (profiles/r01_kernel_stats.csv), for comparison with the live HIP-event time of the stage."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_kernel_stats.csv")
    if not os.path.exists(path) or stage not in STAGE_KERNELS:
        return None
    out = {}
    with open(path) as f:
        for line in f:
            if not line.startswith('"'):
                continue
            name, rest = line[1:].split('",', 1)
            short = name.replace("void ", "").split("(")[0]
            if any(short.startswith(pfx) for pfx in STAGE_KERNELS[stage]):
                out[short] = float(rest.split(",")[1])
    return out or None


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=192, help="fields of view per GPU per step")
    ap.add_argument("--streams", type=int, default=6,
                    help="HIP streams per GPU; the batch is split over them so that the latency-bound flood of one "
                         "part overlaps the bandwidth-bound stages of the other")
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--max-cells", type=int, default=2048,
                    help="row capacity of the per-FOV feature tables (the synthetic FOVs hold ~1,360 nuclei); it sizes "
                         "the per-plate block that the ranks all-gather, and a FOV that exceeds it raises")
    ap.add_argument("--unique", type=int, default=8,
                    help="distinct synthetic FOVs generated per GPU (host-side generation costs ~0.5 s each); the "
                         "batch cycles through them")
    ap.add_argument("--workload", choices=["c3", "c2"], default="c3")
    ap.add_argument("--cpu-fovs", type=int, default=2, help="FOVs timed through the single-thread CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-h2d", action="store_true",
                    help="skip the extra host-fed run (FOVs streamed from pinned host memory over PCIe, N = 1 only)")
    return ap.parse_args()


def cpu_baseline(fovs: np.ndarray, workload: str, n_single: int):
    """Time the CPU oracle (numpy/scipy restatement of the reference's scikit-image path, kind 'port') on a
    bounded sample of the same workload: single thread, then all host cores with the reference's own
    ThreadPoolExecutor mode (R/pipeline.py:145)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import chains

    fn = (lambda f: chains.c3_chain(f)) if workload == "c3" else (lambda f: chains.c2_chain(f[1]))
    n_single = max(1, min(n_single, len(fovs)))
    t0 = time.perf_counter()
    for i in range(n_single):
        fn(fovs[i])
        log(f"cpu baseline: FOV {i + 1}/{n_single} single-thread done")
    t1 = time.perf_counter() - t0
    # this process's CPU share (a 1-GPU box grants 16 cores of a much larger host)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    sample = [fovs[i % len(fovs)] for i in range(cores)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(fn, sample))
    tall = time.perf_counter() - t0
    return {
        "value": n_single / t1,
        "unit": "FOV/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n_single} synthetic 4x{fovs.shape[-1]}^2 FOVs through oracle/chains.py ({workload}), 1 thread",
        "all_cores": {"value": len(sample) / tall, "cores": cores,
                      "sample": f"{len(sample)} FOVs, ThreadPoolExecutor(max_workers={cores})"},
    }


def main():
    args = parse_args()
    # stdout must carry exactly ONE JSON line: RCCL / HIP runtime banners written to fd 1 by native code are
    # sent to stderr instead, and the JSON goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # AMT_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (used to rehearse the N > 1 code)
    distributed = world > 1 or os.environ.get("AMT_BENCH_FORCE_DIST") == "1"

    from arcadia_microscopy_tools_amd import synth
    from arcadia_microscopy_tools_amd.device import Context, set_default_device
    from arcadia_microscopy_tools_amd.segment import FovSegmenter

    torch = dist = None
    if distributed:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # compute runs on the library's own HIP streams (measured: compute placed on a torch-created stream
        # ran ~30 % slower next to a second stream); only the collective uses a torch (side) stream,
        # ordered against the compute streams with amt_stream_wait (plate.PlateGatherer)
        set_default_device(local_rank)
        ctx = Context(local_rank)
    else:
        set_default_device(local_rank)
        ctx = Context(local_rank)

    B, S = args.batch, args.size
    # every rank owns B distinct FOV indices of the plate (weak scaling)
    t0 = time.perf_counter()
    nuniq = max(1, min(args.unique, B))
    uniq = [synth.synth_fov(rank * B + i, size=S) for i in range(nuniq)]
    fovs = np.stack([uniq[i % nuniq] for i in range(B)])
    gen_s = time.perf_counter() - t0
    log(f"rank {rank}: generated {nuniq} distinct synthetic FOVs (batch {B}) in {gen_s:.1f}s; "
        f"device = {ctx.device_name()}")
    d_fovs = ctx.asarray(fovs)
    # split the batch over `streams` contexts (each = one HIP stream + arena); parts run concurrently
    nstreams = max(1, min(args.streams, B))
    bounds = [round(i * B / nstreams) for i in range(nstreams + 1)]
    ctxs = [ctx] + [Context(local_rank) for _ in range(nstreams - 1)]
    parts = [d_fovs[bounds[i]:bounds[i + 1]] for i in range(nstreams)]
    segs = [FovSegmenter(bounds[i + 1] - bounds[i], 4, S, S, ctx=ctxs[i], max_cells=args.max_cells)
            for i in range(nstreams)]
    seg = segs[0]
    packed = None
    if distributed:
        from arcadia_microscopy_tools_amd.plate import PlateTables

        # this rank's blocks of the per-plate feature tables: one plate (= one step of all ranks), one all-gather
        packed = PlateTables(segs, args.steps, torch.device("cuda", local_rank))

    gather = packed is not None and args.workload == "c3" and os.environ.get("AMT_BENCH_NO_GATHER") != "1"

    def step(i=0):
        if gather:
            packed.point(i)
        for sg, part in zip(segs, parts):
            if args.workload == "c3":
                sg.run_c3(part)
            else:
                sg.run_c2(part)
        if gather:
            packed.gather_step(i)  # this plate's single RCCL all-gather; it overlaps the next step's compute

    def sync():
        if distributed:
            torch.cuda.synchronize()
        else:
            for c in ctxs:
                c.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    log("warmup done")
    if distributed:
        dist.barrier()
        sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if gather:
        packed.all_gather()  # nothing left to exchange unless a step was skipped; all of it is inside the timed region
    sync()
    if distributed:
        dist.barrier()
        sync()
    elapsed = time.perf_counter() - t0
    log(f"timed region: {args.steps} steps in {elapsed:.3f}s")
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if gather:  # outside the timed region: what arrived is what was written
        t_all, it_all, nc_all = packed.result()
        mine = nc_all[rank]
        if int(mine.min()) < 0 or int(mine.max()) > args.max_cells or int(nc_all.sum()) <= 0:
            raise RuntimeError("gathered plate tables are inconsistent (cell counts out of range)")
        if world == 1 and not torch.equal(packed.gathered, packed.local):
            raise RuntimeError("single-rank all-gather did not reproduce the local plate blocks")
        log(f"plate tables gathered: {tuple(t_all.shape)} rows, {int(nc_all.sum())} cells over {world} rank(s)")

    # ---- per-stage device times (HIP events on the kernels' own stream), outside the timed region ----
    # one launch of the timed region covers the FOVs of ONE stream: profile that launch size
    PB = bounds[1] - bounds[0]
    prof = FovSegmenter(PB, 4, S, S, ctx=ctx, profile=True, max_cells=args.max_cells)
    d_prof = d_fovs[:PB]
    stage_ms: dict[str, list[float]] = {}
    reps = 3
    for rep in range(reps + 1):
        (prof.run_c3 if args.workload == "c3" else prof.run_c2)(d_prof)
        if rep == 0:  # warm-up: this segmenter's batch is larger than the timed ones, so the arena grows once
            ctx.synchronize()
            continue
        for k, v in prof.times.ms().items():
            stage_ms.setdefault(k, []).append(v)
    stage_avg = {k: float(np.mean(v)) for k, v in stage_ms.items()}
    ncells = prof.ncells.numpy() if args.workload == "c3" else prof.count8.numpy()
    if args.workload == "c3":  # a run whose tables overflowed would have skipped work: refuse to report it
        for sg in segs + [prof]:
            nm = sg.nmarkers.numpy()
            if (nm < 0).any() or (nm > args.max_cells).any():
                raise RuntimeError(f"a field of view produced {int(nm.max())} markers, above --max-cells {args.max_cells}")
    log("stage ms: " + ", ".join(f"{k}={v:.3f}" for k, v in stage_avg.items()))

    # ---- host-fed variant (never `value`): the same steps with every batch arriving from pinned host memory over
    # PCIe, double-buffered on a copy stream so that the transfer of batch i + 1 overlaps the segmentation of i ----
    pcie = None
    if not distributed and not args.no_h2d and args.workload == "c3":
        from arcadia_microscopy_tools_amd.feeder import FovFeeder

        feeder = FovFeeder(fovs.shape, local_rank)
        for slot in range(2):
            feeder.host(slot)[...] = fovs  # the file reader's job; done once here, outside the timed region
        fparts = [[feeder.device(slot)[bounds[i]:bounds[i + 1]] for i in range(nstreams)] for slot in range(2)]
        feeder.submit(0)

        def hstep(i):
            slot = i % 2
            feeder.acquire(slot, ctxs)
            feeder.submit(1 - slot)  # next batch: in flight while this one is segmented
            for sg, part in zip(segs, fparts[slot]):
                sg.run_c3(part)
            feeder.release(slot, ctxs)

        hstep(0)
        sync()
        feeder.copy_ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(1, args.steps + 1):
            hstep(i)
        sync()
        feeder.copy_ctx.synchronize()
        h_elapsed = time.perf_counter() - t0
        pcie = {
            "value": B * args.steps / h_elapsed, "unit": "FOV/s",
            "h2d_GBps": fovs.nbytes * args.steps / h_elapsed / 1e9,
            "note": "batches arrive from page-locked host memory; double-buffered H2D on a copy stream overlaps "
                    "the segmentation; bounded by PCIe (33.55 MB per FOV)",
        }
        log(f"host-fed: {pcie['value']:.0f} FOV/s, H2D {pcie['h2d_GBps']:.1f} GB/s")
        feeder.close()

    if rank == 0:
        npx = PB * S * S  # pixels per launch (one stream's share of the batch)
        dom = max(stage_avg, key=stage_avg.get)
        dom_bytes = STAGE_BYTES_PER_PX[dom] * npx
        achieved = dom_bytes / (stage_avg[dom] * 1e-3) / 1e9
        chain_bytes = sum(STAGE_BYTES_PER_PX[k] for k in stage_avg) * npx
        chain_ms = sum(stage_avg.values())
        # filter + morphology chain of the north_star: Gaussian (10 B/px) + open + close (8 B/px).  The '>' is
        # fused into the packed open/close chain, so that stage's time is charged in full while only the
        # morphology's 8 B/px are credited (conservative).
        fm = [k for k in ("gaussian", "opening", "closing", "threshold_open_close") if k in stage_avg]
        fm_bytes = sum(8 if k == "threshold_open_close" else STAGE_BYTES_PER_PX[k] for k in fm) * npx
        fm_ms = sum(stage_avg[k] for k in fm)
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_of_measured_peak": achieved / 6290.0,  # 6.29 TB/s = measured HBM3E streaming rate (MI355X_MICROARCH.md)
            # PMC bytes of this stage's kernels for a 32-FOV launch, scaled to this run's launch size
            "traffic": (lambda t: None if t is None else t * PB / 32.0)(pmc_traffic_bytes(dom)),
            "algorithmic_bytes_per_launch": dom_bytes, "launch_ms": stage_avg[dom],
            # the stage is one C-ABI call = several kernels (the flood classes run concurrently): their rocprofv3
            # averages from the committed summary, per 32-FOV launch
            "stage_kernels_rocprof_us": rocprof_kernel_us(dom),
            "chain": {"achieved": chain_bytes / (chain_ms * 1e-3) / 1e9, "frac": chain_bytes / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "ms": chain_ms},
            "filter_morphology_chain": {"achieved": fm_bytes / (fm_ms * 1e-3) / 1e9,
                                        "frac": fm_bytes / (fm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": fm_ms},
            "stage_ms": stage_avg,
        }
        total_fovs = world * B * args.steps
        out = {
            "metric": "fields-of-view/sec (4x2048^2 uint16) end-to-end segment+props",
            "value": total_fovs / elapsed,
            "unit": "FOV/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("configs[2]: synthetic 4-channel 2048x2048 uint16 FOVs, DAPI Gaussian+Otsu+open/close+EDT+"
                             "watershed nuclei + morphology and 4-channel intensity regionprops"
                             if args.workload == "c3" else
                             "configs[1]: synthetic 2048x2048 uint16 DAPI plane, Gaussian(2)+Otsu+open/close+CCL"),
                "fovs_per_gpu_per_step": B, "streams_per_gpu": nstreams, "fovs_per_launch": PB,
                "max_cells_per_fov": args.max_cells, "fov_shape": [4, S, S],
                "resident_in_hbm": True,
                "cells_per_fov_mean": float(np.mean(ncells)),
                "feature_table_all_gather": bool(distributed and args.workload == "c3"),
            },
            "roofline": roofline,
            "host_gen_s": gen_s,
        }
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if not args.no_cpu and world == 1:  # the CPU baseline is a rank-0, N = 1 measurement
            out["cpu_baseline"] = cpu_baseline(fovs, args.workload, args.cpu_fovs)
            out["gpu_over_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
