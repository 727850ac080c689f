"""bench.py -- fields-of-view/sec of the end-to-end segment + props hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W --batch B] [--plate 384] [--workload c3|c2|prep|filters]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = the whole config-3 chain (BASELINE.json configs[2]: Gaussian -> Otsu -> threshold -> open ->
close -> EDT -> peak markers -> watershed -> clear_border -> relabel -> morphology + 4-channel intensity
tables) over one batch of synthetic 4 x 2048 x 2048 uint16 fields of view that are ALREADY RESIDENT in HBM when
the timed region starts.

N > 1: one process per GPU.  ``--gpus N`` without a launcher (WORLD_SIZE unset) starts the N ranks itself, as
child processes of a parent that never touches HIP; under ``torch.distributed.run`` the ranks come from the
environment and ``--gpus`` must agree with WORLD_SIZE.  Fields of view are independent: they are sharded with no
collective on the data path, and each step (= one plate) ends with the plate's feature-table exchange over RCCL
(row counts first, then ONE all-gather of the packed rows; arcadia_microscopy_tools_amd/plate.py).
  default        weak scaling: every rank processes --batch fields of view per step (N x 192 per plate)
  --plate 384    strong scaling: BASELINE configs[3], one step = one 384-FOV plate split over the ranks (48 per
                 GPU at N = 8)

Rank 0 prints ONE JSON line (contract in the task statement).  ``--workload prep`` / ``filters`` time the
reference's canonical preprocessing (R/operations.py:57-97,10-54) and the north_star's filter set per stage, with
a roofline per stage; their ``value`` is planes/s and they are supplementary lines, not the headline metric.

The default N = 1 run also times, AFTER the headline and inside the same command (so that the driver's clock covers
them), the supplementary lines ``sublines``: c2, prep, filters, plate48 (the per-GPU share of configs[3]), unique64
(64 distinct FOVs), api (the reference-level batch_segment + cell_properties calls on host arrays) and a8_exact (the
same chain with SURVEY A.8's plain -EDT relief and exact tie handling).  ``--no-sublines`` skips them.

Watershed recipe of the headline: the reference defines no marker recipe.  The timed chain floods the
SEEDED relief (marker pixels are spread first, in raster order: oracle/skops.py:seeded_flood_image, ``seeds_first``),
an ordinary watershed(image, markers, mask) call that cannot tie; SURVEY A.8's plain watershed(-edt, markers, mask)
ties on nearly every plane and is reported separately under ``sublines.a8_exact``.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

# Each context owns up to three HIP streams (main + two fork/join streams for the flood classes) and the
# batch is split over several contexts; ROCm's default of 4 hardware queues per process would multiplex them
# (and with torch + RCCL in the process, 8 measured 14 % slower than 16).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# page-locked result blocks handed out at any time (device._ResultPool): the api line's worker threads hold a few
# int64 label images each (33.5 MB per 2048^2 image); beyond the budget results fall back to pageable arrays
os.environ.setdefault("AMT_RESULT_PINNED_BYTES", str(4 << 30))

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (G/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
# algorithmic bytes per pixel per stage: narrowest dtypes, one read + one write
STAGE_BYTES_PER_PX = {
    "gaussian": 10, "otsu": 8, "gaussian_otsu": 18, "threshold": 9, "opening": 4, "closing": 4, "threshold_open_close": 17, "label8": 5,
    "edt": 9, "peaks": 17, "markers": 5, "watershed": 17, "clear_border": 8, "relabel": 4, "regionprops": 16,
    "watershed_clear_relabel": 25,  # watershed 17 + clear_border / relabel 8, one C-ABI call
    "intensity": 12,
    # reference-level preprocessing (R/operations.py) and the north_star's other filters, per SURVEY.md 8(d)
    "dog": 10, "percentile_f64": 8, "sub_clip": 16, "rescale": 16,
    "median_disk2": 4, "grey_open_disk2": 8, "grey_close_disk2": 8, "tophat_disk7": 14, "grey_erosion_disk2": 4,
}

# kernels (name prefixes in the rocprofv3 output) that make up each stage of the chain
STAGE_KERNELS = {
    "gaussian": ("gauss_lds_kernel", "gauss_fused_kernel", "conv_v8", "conv_h8"),
    "otsu": ("hist_f64_kernel", "otsu_f64_kernel", "minmax_"),
    "gaussian_otsu": ("gauss_lds_kernel", "otsu_f64_kernel", "minmax_"),
    "threshold_open_close": ("pack_gt_kernel", "toc_fused_kernel", "packed_prim_kernel", "unpack_kernel"),
    "edt": ("edt_rows", "edt_cols"),
    "peaks": ("peaks_",),
    "markers": ("sp_",),
    "watershed": ("ccl_", "ws_", "roots_"),
    "watershed_clear_relabel": ("ccl_", "ws_", "roots_", "presence_", "drop_flagged_kernel"),
    "label8": ("ccl_", "roots_", "apply_rank_kernel"),
    "clear_border": ("presence_", "frame_mark_kernel", "drop_flagged_kernel", "map_labels_kernel"),
    "regionprops": ("rp_",),
}
# committed rocprofv3 summaries the `traffic` / `stage_kernels_rocprof_us` fields are read from (NOT measured in
# this run: PMC collection needs separate rocprofv3 passes); the newest tag present wins
PROFILE_TAGS = ("r03", "r02", "r01")


def _profile_path(kind: str):
    for tag in PROFILE_TAGS:
        p = os.path.join(ROOT, "profiles", f"{tag}_{kind}.csv")
        if os.path.exists(p):
            return p
    return None


def pmc_traffic_bytes(stage: str):
    """HBM bytes per 32-FOV launch of one stage from the committed PMC summary (profiles/rNN_hbm_pmc.csv:
    FETCH_SIZE with the gfx950 x2 correction for wide reads + WRITE_SIZE), or None."""
    path = _profile_path("hbm_pmc")
    if path is None or stage not in STAGE_KERNELS:
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


def pmc_total_bytes():
    """HBM bytes of ONE 32-FOV chain execution from the committed PMC summary (its TOTAL row: fetch x2 + write), or None."""
    path = _profile_path("hbm_pmc")
    if path is None:
        return None
    with open(path) as f:
        for line in f:
            if line.startswith("TOTAL"):
                parts = line.rstrip("\n").split(",")
                return (float(parts[3]) + float(parts[4])) * 1e6
    return None


def rocprof_kernel_us(stage: str, launch_fovs: int = 32):
    """Average duration (us per launch) of the kernels of one stage from the committed rocprofv3 summary, for comparison
    with the live HIP-event time of the stage: profiles/rNN_kernel_stats_b48.csv is ONE context alone with 48 FOVs per
    launch and no auxiliary streams (the default run's launch size and settings, what the profiled pass times),
    profiles/rNN_kernel_stats.csv a single context with 32 FOVs per launch and its auxiliary streams."""
    path = _profile_path("kernel_stats_b48") if launch_fovs == 48 else None
    path = path or _profile_path("kernel_stats")
    if path is None or stage not in STAGE_KERNELS:
        return None
    out = {"source": os.path.relpath(path, ROOT)}
    with open(path) as f:
        for line in f:
            if not line.startswith('"'):
                continue
            name, rest = line[1:].split('",', 1)
            short = name.replace("void ", "").split("(")[0]
            if any(short.startswith(pfx) for pfx in STAGE_KERNELS[stage]):
                out[short] = float(rest.split(",")[1])
    return out or None


_FOV_CACHE: dict = {}


def synth_fovs(indices, size):
    """Distinct synthetic FOVs (SURVEY.md 8(d) generator), generated on all host cores and cached for the sublines."""
    from concurrent.futures import ThreadPoolExecutor

    from arcadia_microscopy_tools_amd import synth

    todo = [i for i in indices if (i, size) not in _FOV_CACHE]
    if todo:
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        with ThreadPoolExecutor(max_workers=max(1, min(cores, 16))) as ex:
            for i, f in zip(todo, ex.map(lambda k: synth.synth_fov(k, size=size), todo)):
                _FOV_CACHE[(i, size)] = f
    return [_FOV_CACHE[(i, size)] for i in indices]


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=192, help="fields of view per GPU per step (weak scaling)")
    ap.add_argument("--plate", type=int, default=0,
                    help="strong scaling: fields of view of ONE plate, split over the ranks; a step is one plate "
                         "(384 = BASELINE configs[3]); 0 = weak scaling with --batch per GPU")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams per GPU; the batch is split over them so that the latency-bound flood of one "
                         "part overlaps the bandwidth-bound stages of the others (0 = 6, fewer for small batches)")
    ap.add_argument("--aux-streams", type=int, default=0,
                    help="auxiliary streams per context when several contexts run (0..3; a single context uses 3)")
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--max-cells", type=int, default=2048,
                    help="row capacity of the per-FOV feature tables (the synthetic FOVs hold ~1,360 nuclei); a FOV "
                         "that exceeds it raises")
    ap.add_argument("--unique", type=int, default=8,
                    help="distinct synthetic FOVs generated per GPU (host-side generation costs ~0.5 s each); the "
                         "batch cycles through them")
    ap.add_argument("--workload", choices=["c3", "c2", "prep", "filters", "c5", "api", "a8"], default="c3")
    ap.add_argument("--tiles", type=int, default=8, help="c5: 2 x 1024 x 1024 tiles per step (the network's batch)")
    ap.add_argument("--cpu-fovs", type=int, default=12, help="FOVs timed through the single-thread CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1 without the per-plate feature-table exchange")
    ap.add_argument("--no-h2d", action="store_true",
                    help="skip the extra host-fed run (FOVs streamed from pinned host memory over PCIe, N = 1 only)")
    ap.add_argument("--no-sublines", action="store_true",
                    help="N = 1 default run: skip the supplementary lines (c2, prep, filters, plate48, unique64, api, "
                         "a8_exact) that are otherwise timed after the headline")
    ap.add_argument("--no-deliver", action="store_true",
                    help="N = 1: keep the feature tables on the device (by default every step packs its table and "
                         "copies it to page-locked host memory inside the timed region, two steps lagged)")
    ap.add_argument("--tail-reps", type=int, default=0,
                    help="c3: time the watershed stage of a 32-FOV launch this many times over rotating windows of "
                         "the distinct FOVs and report p50 / p99 (flood time is set by the largest component)")
    return ap.parse_args()


def launch_ranks_if_needed(args):
    """``--gpus N`` without a launcher: start N ranks as fresh child processes BEFORE anything touches HIP (the
    parent only waits for them), one rank per GPU over RCCL, rendezvous on 127.0.0.1."""
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with matching values",
                  file=sys.stderr)
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching: " + " ".join(cmd))
    sys.exit(subprocess.call(cmd))


def _cpu_c3_worker(job):
    """One field of view through the oracle's config-3 chain in a WORKER PROCESS (spawned: no GIL shared with its
    siblings, no HIP state inherited): generates the FOV itself, returns the seconds the chain took."""
    index, size = job
    from arcadia_microscopy_tools_amd import synth
    from oracle import chains

    fov = synth.synth_fov(index, size=size)
    t0 = time.perf_counter()
    chains.c3_chain(fov)
    return time.perf_counter() - t0


def cpu_baseline(fovs: np.ndarray, workload: str, n_single: int):
    """Time the CPU oracle (numpy/scipy restatement of the reference's scikit-image path, kind 'port') on a
    bounded sample of the same workload: single thread, then all host cores with the reference's own
    ThreadPoolExecutor mode (R/pipeline.py:145)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import chains, skops

    if workload == "c3":
        fn = lambda f: chains.c3_chain(f)  # noqa: E731
    elif workload == "c2":
        fn = lambda f: chains.c2_chain(f[1])  # noqa: E731
    elif workload == "prep":
        def fn(f):
            d = skops.difference_of_gaussians(f[1], 0.6, 16.0)
            d = np.clip(d - np.percentile(d, 90), 0, None)
            p = np.percentile(d, (1, 99))
            return skops.rescale_intensity(d, p, (0, 1))
    else:
        def fn(f):
            se2, se7 = skops.disk(2), skops.disk(7)
            return (skops.median(f[1], se2), skops.opening(f[1], se2), skops.closing(f[1], se2),
                    skops.white_tophat(f[1], se7))
    unit = "FOV/s" if workload in ("c3", "c2") else "planes/s"
    n_single = max(1, n_single)  # ~1.1 s per FOV: the default 12 keeps the sample inside the 10-30 s the contract asks for
    t0 = time.perf_counter()
    for i in range(n_single):
        fn(fovs[i % len(fovs)])
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
    procs = None
    if workload == "c3":
        # the same sample with one PROCESS per core: what the host can do when Python's lock is out of the way (the
        # reference itself offers threads only, R/pipeline.py:145, so this is an upper bound on its own CPU path)
        try:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor

            # an executor, not mp.Pool: a worker that dies raises BrokenProcessPool instead of being respawned for ever
            with ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as pool:
                list(pool.map(_cpu_c3_worker, [(0, 256)] * cores, timeout=180))  # start the workers outside the timed part
                t0 = time.perf_counter()
                secs = list(pool.map(_cpu_c3_worker, [(i, fovs.shape[-1]) for i in range(cores)], timeout=300))
                tproc = time.perf_counter() - t0
            # the workers also generate their FOV (~0.5 s each, not part of the chain): the chains ran side by side and
            # the slowest one bounds them
            procs = {"value": cores / max(1e-9, max(secs)), "cores": cores,
                     "sample": f"{cores} FOVs, one spawned process each, slowest chain {max(secs):.2f} s, "
                               f"{tproc:.1f} s with the FOV generation"}
        except Exception as e:  # a box that forbids child processes keeps the thread figure
            procs = {"error": f"{type(e).__name__}: {e}"}
    return {
        "value": n_single / t1,
        "unit": unit,
        "cores": 1,
        "kind": "port",
        "sample": f"{n_single} synthetic 4x{fovs.shape[-1]}^2 FOVs through the oracle ({workload}), 1 thread, "
                  f"{t1:.1f} s",
        "all_cores": {"value": len(sample) / tall, "cores": cores,
                      "sample": f"{len(sample)} FOVs, ThreadPoolExecutor(max_workers={cores}), {tall:.1f} s"},
        "all_cores_processes": procs,
    }


def stage_roofline(stage_avg: dict, npx: int, PB: int):
    """The `roofline` object: dominant stage vs the HBM peak, plus the whole chain and the filter+morphology chain."""
    dom = max(stage_avg, key=stage_avg.get)
    dom_bytes = STAGE_BYTES_PER_PX[dom] * npx
    achieved = dom_bytes / (stage_avg[dom] * 1e-3) / 1e9
    chain_bytes = sum(STAGE_BYTES_PER_PX[k] for k in stage_avg) * npx
    chain_ms = sum(stage_avg.values())
    traffic = pmc_traffic_bytes(dom)
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "frac_of_measured_peak": achieved / 6290.0,  # 6.29 TB/s = measured HBM3E streaming rate (MI355X_MICROARCH.md)
        # PMC bytes of this stage's kernels for a 32-FOV launch from the COMMITTED profile (separate rocprofv3
        # --pmc passes; not measured in this run), scaled to this run's launch size
        "traffic": None if traffic is None else traffic * PB / 32.0,
        "traffic_source": (os.path.relpath(_profile_path("hbm_pmc"), ROOT) + " (committed PMC passes, FETCH_SIZE x2 "
                           "for wide reads + WRITE_SIZE)") if traffic is not None else None,
        # the same stage priced by the bytes its kernels really move: well below `frac` when passes are fused or replaced
        # (round 3: run tables instead of a parent plane) -- what is left of the stage is then latency, not HBM
        "frac_pmc": None if traffic is None else traffic * PB / 32.0 / (stage_avg[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": dom_bytes, "launch_ms": stage_avg[dom],
        # the stage is one C-ABI call = several kernels (the flood classes run concurrently): their rocprofv3
        # averages from the committed summary, per 32-FOV launch
        "stage_kernels_rocprof_us": rocprof_kernel_us(dom, PB),
        "chain": {"achieved": chain_bytes / (chain_ms * 1e-3) / 1e9,
                  "frac": chain_bytes / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": chain_ms},
        "stage_ms": stage_avg,
        "stage_frac": {k: STAGE_BYTES_PER_PX[k] * npx / (v * 1e-3) / 1e9 / HBM_PEAK_GBS for k, v in stage_avg.items()},
    }
    # filter + morphology chain of the north_star: Gaussian (10 B/px) + open + close (8 B/px).  The '>' is
    # fused into the packed open/close chain, so that stage's time is charged in full while only the
    # morphology's 8 B/px are credited (conservative).
    # (With the code path -- stage "gaussian_otsu", two passes of the Gaussian and the histogram in one stage -- the
    # Gaussian's 10 B/px are credited against the time of the whole stage, Otsu included: conservative again.)
    fm = [k for k in ("gaussian", "gaussian_otsu", "opening", "closing", "threshold_open_close") if k in stage_avg]
    if fm:
        fm_bytes = sum(8 if k == "threshold_open_close" else (10 if k == "gaussian_otsu" else STAGE_BYTES_PER_PX[k])
                       for k in fm) * npx
        fm_ms = sum(stage_avg[k] for k in fm)
        roofline["filter_morphology_chain"] = {"achieved": fm_bytes / (fm_ms * 1e-3) / 1e9,
                                               "frac": fm_bytes / (fm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": fm_ms}
    return roofline


# ------------------------------------------------------------------------------------------------------
# reference-level operator chains on B resident planes (supplementary workloads)
# ------------------------------------------------------------------------------------------------------
def run_ops(args):
    from arcadia_microscopy_tools_amd import hipops, synth
    from arcadia_microscopy_tools_amd.device import Context, set_default_device

    set_default_device(0)
    ctx = Context(0)
    B, S = (args.batch if args.batch != 192 else 32), args.size
    t0 = time.perf_counter()
    nuniq = max(1, min(args.unique, B))
    uniq = [synth.synth_fov(i, size=S) for i in range(nuniq)]
    gen_s = time.perf_counter() - t0
    planes = np.stack([uniq[i % nuniq][1] for i in range(B)])  # the DAPI plane of every FOV
    d = ctx.asarray(planes)
    log(f"{B} resident {S}^2 uint16 planes on {ctx.device_name()}")
    f64 = lambda: ctx.empty((B, S, S), np.float64)  # noqa: E731
    u16 = lambda: ctx.empty((B, S, S), np.uint16)  # noqa: E731
    if args.workload == "prep":
        # subtract_background_dog(percentile=90) -> rescale_by_percentile((1, 99)): the notebooks' fluorescence
        # preprocessing (R/operations.py:57-97, 10-54)
        dog, out = f64(), f64()
        lvl = ctx.empty((B, 1), np.float64)
        pr = ctx.empty((B, 2), np.float64)
        stages = [
            ("dog", lambda: hipops.difference_of_gaussians(d, 0.6, 16.0, out=dog)),
            ("percentile_f64", lambda: hipops.percentile(dog, 90.0, out=lvl)),
            ("sub_clip", lambda: hipops.sub_clip0(dog, lvl, out=dog)),
            ("percentile_f64#2", lambda: hipops.percentile(dog, (1.0, 99.0), out=pr)),
            ("rescale", lambda: hipops.rescale(dog, pr, (0.0, 1.0), out=out)),
        ]
        desc = ("reference preprocessing on DAPI planes: subtract_background_dog(0.6, 16, percentile=90) -> "
                "rescale_by_percentile((1, 99)) (R/operations.py:57-97,10-54)")
    else:
        se2, se7 = hipops.disk(2), hipops.disk(7)
        a, b, c = u16(), u16(), u16()
        stages = [
            ("median_disk2", lambda: hipops.median(d, se2, out=a)),
            ("grey_erosion_disk2", lambda: hipops.erosion(d, se2, out=a)),
            ("grey_open_disk2", lambda: hipops.dilation(hipops.erosion(d, se2, out=a), se2, out=b)),
            ("grey_close_disk2", lambda: hipops.erosion(hipops.dilation(d, se2, out=a), se2, out=b)),
            ("tophat_disk7", lambda: hipops.white_tophat(d, se7, out=c)),
        ]
        desc = "north_star filter set on uint16 planes: median disk(2), grey erosion / open / close disk(2), white top-hat disk(7)"

    def step():
        for _, fn in stages:
            fn()

    for _ in range(max(1, args.warmup)):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    timers = {name: ctx.timer() for name, _ in stages}
    acc = {name: [] for name, _ in stages}
    for _ in range(3):
        for name, fn in stages:
            timers[name].start()
            fn()
            timers[name].stop()
        ctx.synchronize()
        for name in acc:
            acc[name].append(timers[name].elapsed_ms())
    stage_avg = {k: float(np.mean(v)) for k, v in acc.items()}
    log("stage ms: " + ", ".join(f"{k}={v:.3f}" for k, v in stage_avg.items()))
    npx = B * S * S
    per_stage = {}
    for k, ms in stage_avg.items():
        bpp = STAGE_BYTES_PER_PX[k.split("#")[0]]
        gbs = bpp * npx / (ms * 1e-3) / 1e9
        per_stage[k] = {"ms": ms, "bytes_per_px": bpp, "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS}
    dom = max(stage_avg, key=stage_avg.get)
    chain_bytes = sum(v["bytes_per_px"] for v in per_stage.values()) * npx
    chain_ms = sum(stage_avg.values())
    if "dog" in per_stage:
        # The DoG is bound by float64 ISSUE, not by HBM: per pixel two passes of the narrow Gaussian
        # (radius 2: 2 + 3 + 2 operations each), two of the wide one (radius 64: 64 + 65 + 64 each) and the subtraction.
        # Bit-exactness forbids fused multiply-add, so the vector unit's ceiling is one float64 operation per lane and
        # clock: 256 CUs x 64 lanes x 2.4 GHz = 39.3 T operations/s (half the 78.6 TFLOP/s that counts an FMA as two;
        # derived from the CU count and the nominal clock -- the guide quotes no float64 figure)
        ops = (2 * 7 + 2 * 193 + 1) * npx
        tops = ops / (stage_avg["dog"] * 1e-3) / 1e12
        per_stage["dog"]["fp64"] = {"operations_per_px": 401, "achieved_Tops": tops, "peak_Tops_without_fma": 39.3,
                                    "frac": tops / 39.3}
    out = {
        "metric": f"planes/sec (2048^2 uint16) through the {args.workload} operator chain", "value": B * args.steps / elapsed,
        "unit": "planes/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if args.workload == "prep" else "u16", "data": "synthetic",
        "config": {"workload": desc, "planes_per_step": B, "plane_shape": [S, S], "resident_in_hbm": True},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": per_stage[dom]["achieved_GBps"], "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": per_stage[dom]["frac"], "traffic": None,
                     "algorithmic_bytes_per_launch": per_stage[dom]["bytes_per_px"] * npx, "launch_ms": stage_avg[dom],
                     "chain": {"achieved": chain_bytes / (chain_ms * 1e-3) / 1e9,
                               "frac": chain_bytes / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": chain_ms},
                     "stages": per_stage},
        "host_gen_s": gen_s,
    }
    if not args.no_cpu:
        fovs = np.stack(uniq)
        out["cpu_baseline"] = cpu_baseline(fovs, args.workload, min(args.cpu_fovs, 4))
        out["gpu_over_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
    return out


# ------------------------------------------------------------------------------------------------------
# config 5: Cellpose-style network forward (bf16, MFMA, PyTorch-ROCm) + HIP flow -> mask post-processing
# ------------------------------------------------------------------------------------------------------
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 (G/MI355X_MICROARCH.md: ~2.5 PF; the 5 PF headline includes 2:1 sparsity)


def run_c5(args, json_fd):
    """One step = forward pass of the RANDOM-WEIGHT architectural stand-in of Cellpose's published U-Net
    (cellpose_hip.make_standin; the real weights are fetched from the network and are unobtainable offline) on
    --tiles tiles of 2 x 1024 x 1024 in bf16, then the HIP post-processing of as many flow fields.  A random network's
    output is not a flow field, so the post-processing runs on SYNTHETIC flow fields of the same shape
    (synth.synthetic_flows: disks with centre-pointing flows, ~1,200 per tile) that stay resident
    on the device; both halves are inside the timed region.  The roofline object is the forward pass against the
    dense bf16 MFMA peak (FLOPs from torch.utils.flop_counter, time from HIP events on torch's stream)."""
    import torch

    from arcadia_microscopy_tools_amd import cellpose_hip as ch
    from arcadia_microscopy_tools_amd import hipops
    from arcadia_microscopy_tools_amd.device import Context, set_default_device
    from arcadia_microscopy_tools_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    set_default_device(local_rank)
    ctx = Context(local_rank)
    T, S = args.tiles, 1024
    # let MIOpen time its convolution algorithms once per shape (the default heuristic pick was 1.5-2x slower here)
    torch.backends.cudnn.benchmark = os.environ.get("AMT_C5_MIOPEN_FIND", "1") == "1"
    net, dt = ch.prepare_network(ch.make_standin(), dev, "bf16")
    x = torch.randn(T, 2, S, S, device=dev, dtype=dt)
    if os.environ.get("AMT_C5_LAYOUT", "nhwc") == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    else:
        net = net.to(memory_format=torch.contiguous_format)
    flops = ch.forward_flops(net, x)
    t0 = time.perf_counter()
    nuniq = max(1, min(args.unique, T, 2))
    syn = [synth.synthetic_flows((S, S), 1200, seed=100 * rank + i) for i in range(nuniq)]
    dP = ctx.asarray(np.stack([syn[i % nuniq][0] for i in range(T)]))
    pr = ctx.asarray(np.stack([syn[i % nuniq][1] for i in range(T)]))
    gen_s = time.perf_counter() - t0
    labels = ctx.empty((T, S, S), np.int32)
    counts = ctx.empty((T,), np.int32)
    log(f"rank {rank}: stand-in network {sum(p.numel() for p in net.parameters()) / 1e6:.2f} M parameters, "
        f"{flops / T / 1e9:.1f} GFLOP per tile; {nuniq} synthetic flow fields in {gen_s:.1f}s")

    # the forward with its elementwise glue fused into HIP passes (cellpose_hip.FusedStandIn; AMT_C5_FUSED=0: eager PyTorch)
    fused = None
    if os.environ.get("AMT_C5_FUSED", "1") == "1" and os.environ.get("AMT_C5_LAYOUT", "nhwc") == "nhwc":
        fused = ch.FusedStandIn(net, Context(local_rank))

    def step():
        with torch.no_grad():
            y = fused(x) if fused is not None else net(x)
        # CellposeModel.eval's own post-processing: 200 flow steps, flow-error filter at the reference's default 0.4
        # (R/model.py:69), size floor 15 and hole filling
        hipops.cellpose_masks(dP, pr, 0.0, 200, out=labels, count=counts, flow_threshold=0.4, fill_holes=True)
        return y

    def sync():
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(max(1, args.warmup)):
        step()
    sync()
    if world > 1:
        dist.barrier()
        sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
        sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # forward pass alone (HIP events on torch's current stream), post-processing alone (HIP events on ctx's stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fwd = []
    for _ in range(5):
        e0.record()
        with torch.no_grad():
            fused(x) if fused is not None else net(x)  # (the fused forward joins torch's current stream when it returns)
        e1.record()
        e1.synchronize()
        fwd.append(e0.elapsed_time(e1))
    fwd_ms = float(np.median(fwd))
    eager_ms = None
    if fused is not None:  # the eager forward beside it: what the fused glue bought
        eg = []
        for _ in range(3):
            e0.record()
            with torch.no_grad():
                net(x)
            e1.record()
            e1.synchronize()
            eg.append(e0.elapsed_time(e1))
        eager_ms = float(np.median(eg))
    tm = ctx.timer()
    post = []
    for _ in range(3):
        tm.start()
        hipops.cellpose_masks(dP, pr, 0.0, 200, out=labels, count=counts, flow_threshold=0.4, fill_holes=True)
        tm.stop()
        ctx.synchronize()
        post.append(tm.elapsed_ms())
    post_ms = float(np.median(post))
    nmask = counts.numpy()
    if (nmask < 0).any():
        raise RuntimeError("the post-processing ran out of seed capacity")
    if rank == 0:
        achieved = flops / (fwd_ms * 1e-3) / 1e12
        out = {
            "metric": "tiles/sec (2x1024^2) Cellpose-style forward (bf16) + HIP flow->mask post-processing",
            "value": world * T * args.steps / elapsed, "unit": "tiles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": "configs[4]: U-Net forward on 2-channel 1024x1024 tiles (bf16 MFMA, PyTorch-ROCm) + flow->mask "
                            "post-processing in HIP",
                "network": "random-weight stand-in of Cellpose's published residual U-Net (6.6 M parameters); real "
                           "weights are unobtainable offline -- throughput only, no accuracy claim",
                "tiles_per_gpu_per_step": T, "tile_shape": [2, S, S], "niter": 200, "flow_threshold": 0.4,
                "postprocessing": "follow flows, seeds, size ceiling, flow-error filter (float64 heat diffusion per mask), "
                                  "size floor 15, hole filling (amt_cellpose_masks_ex)",
                "postprocessing_input": "synthetic flow fields (~1,200 disks per tile), resident on the device",
                "masks_per_tile_mean": float(nmask.mean()),
            },
            "roofline": {"bound": "mfma",
                         "kernel": "network forward (MIOpen convolutions via PyTorch-ROCm" +
                                   ("; batch norm / ReLU / additions / upsampling fused into amt_nn_affine_act_bf16 passes)"
                                    if fused is not None else ", eager)"),
                         "eager_forward_ms": eager_ms,
                         "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": None,
                         "flops_per_launch": flops, "launch_ms": fwd_ms,
                         "postprocessing_ms_per_step": post_ms, "postprocessing_ms_per_tile": post_ms / T},
            "host_gen_s": gen_s,
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------
# the reference-level calls on HOST arrays: batch_segment -> SegmentationMask(...).cell_properties
# ------------------------------------------------------------------------------------------------------
def run_api(args):
    """What a drop-in user runs (R/model.py:217-290 + R/masks.py:247-328): host numpy FOVs through
    ``SegmentationModel(backend="classical").batch_segment`` (int64 label images back on the host), then one
    ``SegmentationMask(labels, {channel: plane x 4}).cell_properties`` per FOV (labels and the four intensity planes go
    back to the device, the feature table comes home).  Worker threads, one context (= HIP stream + page-locked
    staging) each, as the reference's own parallel mode (R/pipeline.py:139-149).  One step = ``--batch`` FOVs (48 when
    left at the default)."""
    from concurrent.futures import ThreadPoolExecutor

    from arcadia_microscopy_tools_amd.channels import BRIGHTFIELD, DAPI, FITC, TRITC
    from arcadia_microscopy_tools_amd.device import set_default_device
    from arcadia_microscopy_tools_amd.masks import SegmentationMask
    from arcadia_microscopy_tools_amd.model import SegmentationModel

    set_default_device(0)
    S = args.size
    B = args.batch if args.batch != 192 else 48
    workers = int(os.environ.get("AMT_API_WORKERS", "4"))
    per_call = int(os.environ.get("AMT_API_CHUNK", "2"))
    t0 = time.perf_counter()
    nuniq = max(1, min(args.unique, B))
    uniq = synth_fovs(list(range(nuniq)), S)
    gen_s = time.perf_counter() - t0
    fovs = [uniq[i % nuniq] for i in range(B)]
    chans = (BRIGHTFIELD, DAPI, FITC, TRITC)
    model = SegmentationModel(backend="classical")
    bus_rate = 53e9  # measured host-link rate of this box class (tools/xfer_probe.py; PCIe Gen5 x16: 63 GB/s nominal)

    def work(chunk):
        masks = model.batch_segment([f[1] for f in chunk], batch_size=len(chunk), show_progress=False)
        out = []
        for f, m in zip(chunk, masks):
            sm = SegmentationMask(m, {c: f[i] for i, c in enumerate(chans)})
            out.append((0, m if f is fovs[0] or f is fovs[min(B - 1, per_call + 1)] else None, sm.cell_properties))
        return out

    chunks = [fovs[i:i + per_call] for i in range(0, B, per_call)]
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for _ in range(max(1, args.warmup)):
            res = [r for part in ex.map(work, chunks) for r in part]
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = [r for part in ex.map(work, chunks) for r in part]
        elapsed = time.perf_counter() - t0
    # the batch results equal the per-image calls (segment(), then the same SegmentationMask)
    equal = True
    for k in (0, min(B - 1, per_call + 1)):
        m1 = model.segment(fovs[k][1])
        p1 = SegmentationMask(m1, {c: fovs[k][i] for i, c in enumerate(chans)}).cell_properties
        equal = (equal and res[k][1] is not None and np.array_equal(m1, res[k][1])
                 and all(np.array_equal(p1[c], res[k][2][c], equal_nan=True) for c in p1))
    if not equal:
        raise RuntimeError("batch_segment + cell_properties differ from the per-image calls")
    # ---- the same results through the one-call form (an addition to the reference's interface): every image crosses
    # the bus once, labels + rows stay on the device, only the rows of the cells that exist come home ----
    mask_chunk = int(os.environ.get("AMT_API_MASK_CHUNK", "8"))
    mworkers = int(os.environ.get("AMT_API_MASK_WORKERS", "2"))
    MB = int(os.environ.get("AMT_API_MASK_FOVS", "192"))  # FOVs per step of this form (a plate's worth per call)
    mfovs = [uniq[i % nuniq] for i in range(MB)]
    share = -(-MB // mworkers)
    shares = [mfovs[i:i + share] for i in range(0, MB, share)]

    def mwork(part):
        masks = model.batch_masks(part, chans, nuclear=DAPI, batch_size=mask_chunk)
        return [(m, m.cell_properties) for m in masks]

    with ThreadPoolExecutor(max_workers=mworkers) as ex:
        for _ in range(max(1, args.warmup)):
            mres = [r for part in ex.map(mwork, shares) for r in part]
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mres = [r for part in ex.map(mwork, shares) for r in part]
        m_elapsed = time.perf_counter() - t0
    for k in (0, min(MB - 1, per_call + 1)):
        m1 = model.segment(mfovs[k][1])
        sm = SegmentationMask(m1, {c: mfovs[k][i] for i, c in enumerate(chans)})
        p1 = sm.cell_properties
        if not (np.array_equal(sm.label_image, mres[k][0].label_image)
                and list(p1) == list(mres[k][1])
                and all(np.array_equal(p1[c], mres[k][1][c], equal_nan=True) for c in p1)):
            raise RuntimeError("batch_masks differs from segment + SegmentationMask")
    one_call = {
        "value": MB * args.steps / m_elapsed, "unit": "FOV/s", "ms_per_step": m_elapsed / args.steps * 1e3,
        "fovs_per_step": MB,
        "workload": "host numpy FOVs -> SegmentationModel(backend='classical').batch_masks(images, channels, "
                    "nuclear=DAPI) -> SegmentationMask objects (labels on the device, downloaded on access) -> "
                    ".cell_properties of every mask on the host",
        "worker_threads": mworkers, "images_per_chunk": mask_chunk,
        "pcie_bound_fov_per_s": bus_rate / (S * S * 8), "h2d_bytes_per_fov": S * S * 8,
        "results_equal_two_calls": True,
    }
    del mres
    # bytes that must cross the bus per FOV on this API: DAPI plane up, int64 labels down, labels (narrowed to int32 on
    # their way into the staging buffer) + four uint16 planes up
    h2d = S * S * (2 + 4 + 4 * 2)
    d2h = S * S * 8
    bus = bus_rate
    return {
        "metric": "fields-of-view/sec through the reference-level API on host arrays (batch_segment + cell_properties)",
        "value": B * args.steps / elapsed, "unit": "FOV/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "host numpy FOVs -> SegmentationModel(backend='classical').batch_segment -> int64 labels on "
                               "the host -> SegmentationMask(labels, 4 channels).cell_properties -> feature dict on the host "
                               "(R/model.py:217-290, R/masks.py:247-328)",
                   "fovs_per_step": B, "worker_threads": workers, "images_per_batch_segment_call": per_call,
                   "fov_shape": [4, S, S], "cells_per_fov_mean": float(np.mean([len(r[2]["label"]) for r in res]))},
        "pcie_bound": {"h2d_bytes_per_fov": h2d, "d2h_bytes_per_fov": d2h, "assumed_GBps_per_direction": bus / 1e9,
                       "fov_per_s": bus / max(h2d, d2h),
                       "note": "labels travel host <-> device twice because the API hands int64 numpy label images from "
                               "batch_segment to SegmentationMask; the resident-FOV headline needs 33.5 MB per FOV one way"},
        # what the host's memory system moves per FOV on this API (reads + writes of the arrays the calls touch, DMA
        # traffic included): DAPI staging 25 MB, label image landing 33.5, its validation pass 33.5, its narrowing upload
        # 67, four planes staged and sent 100 -- the worker threads saturate host memory before they saturate the bus
        "host_memory_traffic_bytes_per_fov": S * S * (6 + 8 + 8 + 16 + 24),
        "results_equal_per_image_calls": True, "host_gen_s": gen_s,
        "one_call": one_call,
    }


# ------------------------------------------------------------------------------------------------------
# SURVEY A.8 as written: watershed(-edt, markers, mask) with exact tie handling
# ------------------------------------------------------------------------------------------------------
def run_a8(args):
    """The config-3 chain with the PLAIN -EDT relief and ties='exact': planes in which two markers of
    one component carry the same value (on an EDT relief: nearly all) are re-flooded by the sequential emulation of
    scikit-image's single heap (bit-identical; one wave per plane).  A short run: ``--batch`` FOVs (48 when left at the
    default) per step."""
    from arcadia_microscopy_tools_amd.device import Context, set_default_device
    from arcadia_microscopy_tools_amd.segment import FovSegmenter

    set_default_device(0)
    ctx = Context(0)
    S = args.size
    B = args.batch if args.batch != 192 else 48  # one wave per plane does the heap: planes are the only parallelism
    nuniq = max(1, min(args.unique, B))
    uniq = synth_fovs(list(range(nuniq)), S)
    d = ctx.asarray(np.stack([uniq[i % nuniq] for i in range(B)]))
    seg = FovSegmenter(B, 4, S, S, ctx=ctx, max_cells=args.max_cells, relief="plain", ties="exact")
    for _ in range(max(1, args.warmup)):
        seg.run_c3(d)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        seg.run_c3(d)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    tied = seg.tied.numpy()
    prof = FovSegmenter(B, 4, S, S, ctx=ctx, max_cells=args.max_cells, relief="plain", ties="exact", profile=True)
    prof.run_c3(d)
    ctx.synchronize()
    stage_ms = prof.times.ms()
    return {
        "metric": "fields-of-view/sec (4x2048^2 uint16), config 3 with SURVEY A.8's plain -EDT relief, ties exact",
        "value": B * args.steps / elapsed, "unit": "FOV/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config 3 with watershed(-edt, markers, mask) as SURVEY A.8 writes it (seeds_first=False, "
                               "ties='exact'): tied planes take the sequential single-heap emulation",
                   "fovs_per_step": B, "fov_shape": [4, S, S]},
        "tied_plane_fraction": float((tied != 0).mean()),
        "roofline": {"kernel": "watershed", "stage_ms": stage_ms,
                     "launch_ms": stage_ms.get("watershed"), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                     "achieved": STAGE_BYTES_PER_PX["watershed"] * B * S * S / (stage_ms["watershed"] * 1e-3) / 1e9,
                     "frac": STAGE_BYTES_PER_PX["watershed"] * B * S * S / (stage_ms["watershed"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "note": "the CPU (scikit-image's own watershed) floods one such plane in 0.3-0.5 s (BASELINE.md section 2)",
    }


# ------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    launch_ranks_if_needed(args)
    # stdout must carry exactly ONE JSON line: RCCL / HIP runtime banners written to fd 1 by native code are
    # sent to stderr instead, and the JSON goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.workload in ("prep", "filters", "api", "a8"):
        if world > 1:
            print(f"bench.py: --workload {args.workload} is a single-GPU line", file=sys.stderr)
            sys.exit(2)
        out = {"prep": run_ops, "filters": run_ops, "api": run_api, "a8": run_a8}[args.workload](args)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        return
    if args.workload == "c5":
        return run_c5(args, json_fd)
    out = run_chain(args)
    default_run = (world == 1 and args.workload == "c3" and args.plate == 0 and not args.no_sublines
                   and os.environ.get("AMT_BENCH_FORCE_DIST") != "1")
    if out is not None and default_run:
        out["sublines"] = run_sublines(args)
    if out is not None:
        os.write(json_fd, (json.dumps(out) + "\n").encode())


def _sub_args(args, **kw):
    d = dict(vars(args))
    d.update(kw)
    return argparse.Namespace(**d)


def _brief(line: dict) -> dict:
    """What a supplementary line keeps of a full bench line."""
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline",
            "pcie_bound", "results_equal_per_image_calls", "note", "tied_plane_fraction", "delivered_to_host",
            "one_call")
    out = {k: line[k] for k in keep if k in line}
    r = out.get("roofline")
    if isinstance(r, dict):
        out["roofline"] = {k: r[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "launch_ms", "chain", "stage_ms",
                                             "stage_frac", "stages", "filter_morphology_chain", "watershed_tail",
                                             "end_to_end") if k in r}
    return out


def run_sublines(args) -> dict:
    """The supplementary lines of the default N = 1 run, a few steps each, timed inside this same command."""
    import gc

    subs = {}

    def attempt(name, fn):
        t0 = time.perf_counter()
        try:
            subs[name] = _brief(fn())
            subs[name]["wall_s"] = time.perf_counter() - t0
            log(f"subline {name}: {subs[name]['value']:.1f} {subs[name]['unit']} ({subs[name]['wall_s']:.1f} s)")
            if "one_call" in subs[name]:
                log(f"subline {name}, one-call form (batch_masks): {subs[name]['one_call']['value']:.1f} FOV/s")
        except Exception as e:  # a supplementary line must not take the headline down with it
            subs[name] = {"error": f"{type(e).__name__}: {e}"}
            log(f"subline {name} FAILED: {subs[name]['error']}")
        gc.collect()

    quick = dict(no_cpu=True, no_h2d=True, no_sublines=True, tail_reps=0)
    attempt("c2", lambda: run_chain(_sub_args(args, workload="c2", steps=10, warmup=2, **quick)))
    def in_child(*argv):
        # a fresh process, as a user's script would be: no heap this command has churned for a minute, no idle contexts
        # and streams of the lines before (measured: the 48-FOV plate 9.8 k FOV/s in here, 10.6-11.1 k on its own)
        cmd = [sys.executable, os.path.abspath(__file__), *argv, "--size", str(args.size), "--unique", str(min(args.unique, 8))]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"child exited with {r.returncode}: {r.stderr.decode()[-300:]}")
        return json.loads(lines[-1])

    attempt("plate48", lambda: in_child("--plate", "48", "--steps", "60", "--warmup", "5", "--no-sublines", "--no-cpu",
                                        "--no-h2d"))
    attempt("unique64", lambda: run_chain(_sub_args(args, unique=64, steps=5, warmup=1, **dict(quick, tail_reps=24))))
    attempt("prep", lambda: run_ops(_sub_args(args, workload="prep", steps=10, warmup=2, no_cpu=True)))
    attempt("filters", lambda: run_ops(_sub_args(args, workload="filters", steps=10, warmup=2, no_cpu=True)))
    # the reference-level calls are bound by the HOST (page-locked pools, copy threads, allocator state): a child too
    attempt("api", lambda: in_child("--workload", "api", "--steps", "5", "--warmup", "2"))
    attempt("a8_exact", lambda: run_a8(_sub_args(args, steps=1, warmup=1)))
    return subs


def run_chain(args):
    """One line of the config-2 / config-3 chain on resident FOVs (the headline, plate48, unique64, c2)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # AMT_BENCH_FORCE_DIST=1 exercises the collective path with a single rank; AMT_BENCH_BACKEND=gloo +
    # AMT_BENCH_SHARE_GPU=1 rehearse N ranks on ONE card (RCCL refuses two ranks on one device)
    distributed = world > 1 or os.environ.get("AMT_BENCH_FORCE_DIST") == "1"
    backend = os.environ.get("AMT_BENCH_BACKEND", "nccl")

    from arcadia_microscopy_tools_amd import _hip, synth
    from arcadia_microscopy_tools_amd.device import Context, set_default_device
    from arcadia_microscopy_tools_amd.plate import shard_indices
    from arcadia_microscopy_tools_amd.segment import FovSegmenter

    device = local_rank
    if os.environ.get("AMT_BENCH_SHARE_GPU") == "1":
        device = local_rank % max(1, _hip.load_library().amt_device_count())
    torch = dist = None
    if distributed:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    # compute runs on the library's own HIP streams (measured: compute placed on a torch-created stream ran ~30 %
    # slower next to a second stream); only the collectives use a torch (side) stream, ordered against the compute
    # streams with amt_stream_wait (plate.PlateTables)
    set_default_device(device)
    ctx = Context(device)

    S = args.size
    if args.plate > 0:  # strong scaling: this rank's contiguous block of the plate
        mine = shard_indices(args.plate, rank, world)
        B = len(mine)
        fov0 = mine[0] if mine else 0
        max_B = -(-args.plate // world)
    else:               # weak scaling: every rank owns B FOV indices of an N x B plate
        B = args.batch
        fov0 = rank * B
        max_B = B
    if B <= 0:
        raise SystemExit(f"rank {rank} has no field of view to process (plate {args.plate} over {world} ranks)")
    t0 = time.perf_counter()
    nuniq = max(1, min(args.unique, B))
    uniq = synth_fovs([fov0 + i for i in range(nuniq)], S)
    fovs = np.stack([uniq[i % nuniq] for i in range(B)])
    gen_s = time.perf_counter() - t0
    log(f"rank {rank}: generated {nuniq} distinct synthetic FOVs (batch {B}) in {gen_s:.1f}s; "
        f"device = {ctx.device_name()}")
    d_fovs = ctx.asarray(fovs)
    # split the batch over `streams` contexts (each = one HIP stream + arena); parts run concurrently
    nstreams = args.streams if args.streams > 0 else max(1, min(4, B // 12))
    nstreams = max(1, min(nstreams, B))
    bounds = [round(i * B / nstreams) for i in range(nstreams + 1)]
    ctxs = [ctx] + [Context(device) for _ in range(nstreams - 1)]
    if nstreams > 1 and os.environ.get("AMT_FORK") is None:
        # several contexts run side by side: the overlap comes from them, not from auxiliary streams inside a call
        # (measured: 4 contexts x 48 FOVs 11.1 k FOV/s without, 6 x 32 with auxiliary streams 10.2 k)
        for c in ctxs:
            c.set_fork(args.aux_streams)
    parts = [d_fovs[bounds[i]:bounds[i + 1]] for i in range(nstreams)]
    segs = [FovSegmenter(bounds[i + 1] - bounds[i], 4, S, S, ctx=ctxs[i], max_cells=args.max_cells,
                         bin_plane=os.environ.get("AMT_BENCH_NO_BINS") != "1",
                         low_traffic=os.environ.get("AMT_BENCH_LOW_TRAFFIC") == "1")
            for i in range(nstreams)]
    packed = None
    gather = distributed and args.workload == "c3" and not args.no_gather and os.environ.get("AMT_BENCH_NO_GATHER") != "1"
    if gather:
        from arcadia_microscopy_tools_amd.plate import PlateTables

        # this rank's staging ring + packed row blocks of the per-plate feature tables
        packed = PlateTables(segs, torch.device("cuda", device), cap_fovs=max_B, keep=2)

    # N = 1: there is no exchange, but the plate's feature table still has to reach the host -- every step packs
    # its table and copies the rows that exist to page-locked host memory (plate.HostTables, two steps lagged)
    deliver = (not distributed) and args.workload == "c3" and not args.no_deliver
    host_tables = None
    if deliver:
        from arcadia_microscopy_tools_amd.plate import HostTables

        host_tables = HostTables(segs, slots=4, lag=2)

    def step(i=0):
        if gather:
            packed.point(i)
        if deliver:
            host_tables.point(i)
        for sg, part in zip(segs, parts):
            if args.workload == "c3":
                sg.run_c3(part)
            else:
                sg.run_c2(part)
        if gather:
            # this plate's exchange: row counts now, the ONE all-gather of its rows when the next step has been
            # enqueued (it overlaps that step's compute)
            packed.gather_step(i, fov_index0=fov0)
        if deliver:
            host_tables.deliver_step(i, fov_index0=fov0)

    def sync():
        if distributed:
            torch.cuda.synchronize()
        else:
            for c in ctxs:
                c.synchronize()

    def barrier():
        if distributed:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    if gather:
        packed.all_gather()
    if deliver:
        host_tables.flush()
    sync()
    log("warmup done")
    n_warm_blocks = packed.exchange.n_finished if gather else 0
    barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if gather:
        packed.all_gather()  # the last plate's rows: every exchange is inside the timed region
    if deliver:
        host_tables.flush()  # the last step's rows: every copy is inside the timed region
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    log(f"timed region: {args.steps} steps in {elapsed:.3f}s")
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    exchange_info = None
    if gather:  # outside the timed region: what arrived is what was written
        res = packed.result()
        if packed.exchange.n_finished - n_warm_blocks != args.steps:
            raise RuntimeError(f"{packed.exchange.n_finished - n_warm_blocks} plates were exchanged, expected {args.steps}")
        rows_last, counts_last = res[-1]
        ncols = packed.ncols
        if rows_last.shape != (sum(counts_last), ncols) or min(counts_last) <= 0:
            raise RuntimeError(f"gathered plate table is inconsistent: {rows_last.shape}, counts {counts_last}")
        # this rank's own rows inside the gathered table are the rows its pack kernel wrote
        own = packed.rows[(args.warmup + args.steps - 1) % packed.slots][: counts_last[rank]].cpu().numpy()
        lo = sum(counts_last[:rank])
        if not np.array_equal(rows_last[lo: lo + counts_last[rank]], own):
            raise RuntimeError("the gathered plate table does not contain this rank's packed rows")
        fcol = rows_last[:, 0]
        expect_fovs = world * B if args.plate <= 0 else args.plate
        if len(np.unique(fcol)) != expect_fovs:
            raise RuntimeError(f"plate table holds {len(np.unique(fcol))} fields of view, expected {expect_fovs}")
        exchange_info = {
            "protocol": "row counts all-gather, then one all-gather of the packed rows (max count over ranks)",
            "rows_per_plate": int(sum(counts_last)), "row_bytes": ncols * 8,
            "bytes_sent_per_rank_per_plate": int(max(counts_last)) * ncols * 8,
            "dense_block_bytes_per_rank": B * args.max_cells * (ncols - 2) * 8,
            "backend": "rccl" if backend == "nccl" else backend,
        }
        log(f"plate tables exchanged: {rows_last.shape} rows in the last plate, counts {counts_last}")

    delivered = None
    if deliver:  # outside the timed region: what arrived on the host is what the segmenters computed
        last = args.warmup + args.steps - 1
        rows = host_tables.rows_of(last)
        ncells_all = np.concatenate([sg.ncells.numpy() for sg in segs])
        if rows.shape[0] != int(ncells_all.sum()) or len(np.unique(rows[:, 0])) != B:
            raise RuntimeError(f"delivered table is inconsistent: {rows.shape[0]} rows for {int(ncells_all.sum())} cells")
        t_dev = np.concatenate([sg.table.numpy()[b, : ncells_all[off + b]] for off, sg in
                                zip(np.cumsum([0] + [g.B for g in segs[:-1]]), segs) for b in range(sg.B)])
        if not np.array_equal(rows[:, 2: 2 + t_dev.shape[1]], t_dev, equal_nan=True):
            raise RuntimeError("the delivered feature table differs from the device tables")
        delivered = {"what": "packed per-cell rows (fov index, label, 14 morphology + 4 x 4 intensity columns) copied to "
                             "page-locked host memory inside the timed region, two steps lagged; label images stay in HBM",
                     "rows_per_step": int(rows.shape[0]), "row_bytes": host_tables.ncols * 8,
                     "bytes_per_step": int(rows.shape[0]) * host_tables.ncols * 8}
        log(f"feature tables delivered to the host: {rows.shape[0]} rows per step")

    # ---- per-stage device times (HIP events on the kernels' own stream), outside the timed region ----
    # one launch of the timed region covers the FOVs of ONE stream: profile that launch size
    PB = bounds[1] - bounds[0]
    prof = FovSegmenter(PB, 4, S, S, ctx=ctx, profile=True, max_cells=args.max_cells,
                        bin_plane=os.environ.get("AMT_BENCH_NO_BINS") != "1",
                        low_traffic=os.environ.get("AMT_BENCH_LOW_TRAFFIC") == "1")
    d_prof = d_fovs[:PB]
    stage_ms: dict[str, list[float]] = {}
    reps = 3
    for rep in range(reps + 1):
        (prof.run_c3 if args.workload == "c3" else prof.run_c2)(d_prof)
        if rep == 0:  # warm-up: this segmenter's arena grows once
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

    # ---- flood tail: the watershed stage over rotating windows of the distinct FOVs (c3 only) ----
    tail = None
    if args.workload == "c3" and args.tail_reps > 0 and rank == 0:
        ws_ms = []
        for rep in range(args.tail_reps):
            start = (rep * max(1, PB // 4)) % B
            idx = [(start + j) % B for j in range(PB)]
            if idx == list(range(idx[0], idx[0] + PB)):
                window = d_fovs[idx[0]: idx[0] + PB]
            else:  # wrap-around window: gather a copy (outside any timed region)
                window = ctx.asarray(fovs[idx])
            prof.run_c3(window)
            ctx.synchronize()
            ws_ms.append(prof.times.ms()["watershed_clear_relabel"])
        ws = np.sort(np.array(ws_ms))
        tail = {"launch_fovs": PB, "distinct_fovs": nuniq, "reps": len(ws_ms), "p50_ms": float(np.percentile(ws, 50)),
                "p99_ms": float(np.percentile(ws, 99)), "min_ms": float(ws[0]), "max_ms": float(ws[-1])}
        log(f"watershed stage over {len(ws_ms)} windows: p50 {tail['p50_ms']:.3f} ms, p99 {tail['p99_ms']:.3f} ms")

    # ---- host-fed variant (never `value`): the same steps with every batch arriving from pinned host memory over
    # PCIe, double-buffered on a copy stream so that the transfer of batch i + 1 overlaps the segmentation of i ----
    pcie = None
    if not distributed and not args.no_h2d and args.workload == "c3":
        from arcadia_microscopy_tools_amd.feeder import FovFeeder

        feeder = FovFeeder(fovs.shape, device)
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

    out = None
    if rank == 0:
        npx = PB * S * S  # pixels per launch (one stream's share of the batch)
        roofline = stage_roofline(stage_avg, npx, PB)
        if tail is not None:
            roofline["watershed_tail"] = tail
        fovs_per_step = args.plate if args.plate > 0 else world * B
        total_fovs = fovs_per_step * args.steps
        out = {
            "metric": "fields-of-view/sec (4x2048^2 uint16) end-to-end segment+props",
            "value": total_fovs / elapsed,
            "unit": "FOV/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.plate > 0 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("configs[2]: synthetic 4-channel 2048x2048 uint16 FOVs, DAPI Gaussian+Otsu+open/close+EDT+"
                             "peak markers+watershed nuclei (SEEDED relief, seeds_first: marker pixels spread first in "
                             "raster order, cannot tie; SURVEY A.8's plain -EDT relief with exact ties is reported "
                             "under sublines.a8_exact) + clear_border/relabel + morphology and 4-channel intensity "
                             "regionprops"
                             if args.workload == "c3" else
                             "configs[1]: synthetic 2048x2048 uint16 DAPI plane, Gaussian(2)+Otsu+open/close+CCL"),
                "mode": (f"configs[3]: one step = one {args.plate}-FOV plate sharded over {world} GPU(s)"
                         if args.plate > 0 else f"weak scaling, {B} FOVs per GPU per step"),
                "fovs_per_step": fovs_per_step, "fovs_per_gpu_per_step": B, "streams_per_gpu": nstreams,
                "fovs_per_launch": PB, "max_cells_per_fov": args.max_cells, "fov_shape": [4, S, S],
                "distinct_fovs_per_gpu": nuniq, "resident_in_hbm": True,
                "cells_per_fov_mean": float(np.mean(ncells)),
                "feature_table_exchange": exchange_info,
            },
            "roofline": roofline,
            "host_gen_s": gen_s,
        }
        if args.workload == "c3":
            # the whole job against the chain's roofline (448.8 MB algorithmic per FOV -> 17.8 k FOV/s at
            # 8 TB/s), and against the bytes the kernels really move (committed PMC passes, per FOV)
            alg_mb = 107 * S * S / 1e6  # C2's 40 B/px + 67 B/px for EDT ... intensity props
            roof_fovs = HBM_PEAK_GBS * 1e3 / alg_mb
            pmc = pmc_total_bytes()
            e2e = {"roofline_fovs_per_s_per_gpu": roof_fovs, "algorithmic_MB_per_fov": alg_mb,
                   "frac": out["value"] / world / roof_fovs,
                   "frac_of_measured_peak": out["value"] / world / (6290.0 * 1e3 / alg_mb)}
            if pmc is not None and S == 2048:
                e2e["pmc_MB_per_fov"] = pmc / 32.0 / 1e6
                e2e["pmc_GBps"] = out["value"] / world * pmc / 32.0 / 1e9
                e2e["frac_pmc"] = e2e["pmc_GBps"] / HBM_PEAK_GBS
                e2e["pmc_source"] = os.path.relpath(_profile_path("hbm_pmc"), ROOT) + " (TOTAL row, per 32-FOV chain)"
            roofline["end_to_end"] = e2e
        if delivered is not None:
            out["delivered_to_host"] = delivered
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if not args.no_cpu and world == 1:  # the CPU baseline is a rank-0, N = 1 measurement
            out["cpu_baseline"] = cpu_baseline(np.stack(uniq), args.workload, args.cpu_fovs)
            out["gpu_over_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
            out["gpu_over_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["all_cores"]["value"]
            pr = out["cpu_baseline"].get("all_cores_processes")
            if pr and "value" in pr:
                out["gpu_over_cpu_all_cores_processes"] = out["value"] / pr["value"]
    if host_tables is not None:
        host_tables.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
