#!/usr/bin/env python3
"""Benchmark of the per-frame pixel chain (BASELINE.json).  Default = the headline configuration:
frames/s of the full fused chain (5x5 Gaussian on HSV, HSV inRange, 3x3 erode + dilate, moments) on
1920x1200 BGR8 frames, a device-resident batch of 256 frames per GPU per step, plus the achieved HBM
GB/s of the chain kernel against the MI355X roofline and the CPU port of the same chain timed on the
host's cores.

    python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4|c5|pp|pe]

--gpus N > 1 without WORLD_SIZE in the environment: this process (which never touches the GPU) starts N
rank processes itself, one per GPU, and relays rank 0's JSON line.  Under torch.distributed.run
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* come from the launcher and nothing is spawned.

One process per GPU; frames are sharded across ranks with NO data-path collective (the path has no
cross-frame step).  The only inter-rank traffic is the timing barrier / max-reduction and the host-side
gather of the per-frame moments (control plane, gloo).

Configurations (BASELINE.json `configs`):
  c1  configs[0]  640x480 BGR8, HSV inRange + 3x3 erode + 3x3 dilate + centroid, no blur: the reference's own CPU-runnable
                  case.  The CPU leg (`cpu_baseline`) is this configuration's subject; the GPU leg is timed beside it
  c2  configs[1]  1920x1200 BGR8, 256 frames per GPU per step, G5 -> inRange -> E3 -> D3 -> moments        (weak scaling)
  c3  configs[2]  2448x2048 BayerRG8, demosaic inside the chain kernel, 64 frames per GPU per step; plus the
                  streamed leg (host frames -> pinned slots -> H2D -> chain), reported beside it, never as `value`
  c4  configs[3]  3840x2160 BGR8, a 4096-frame stream per step cut into contiguous blocks by shard_bounds,
                  per-frame centroids gathered on rank 0                                                      (strong scaling)
  c5  configs[4]  7680x4320 BGR8, 15x15 Gaussian + 9x9 elliptical erode/dilate + 8 ranges, 4 frames per GPU per step
  pp / pe         the reference nodes' own chains (polygon_path; pose_estimate with its commented blur + morphOps) at 1920x1200
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PREROLL_S = 0.15  # untimed steady-state pre-roll before the warmup steps (GPU clock ramp)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# wave64 VALU, the machine's peak: 256 CUs x 4 SIMD-32 x 32 lanes per cycle at 2.4 GHz (MI355X_MICROARCH.md: a wave64
# instruction issues over 2 cycles; = the 157.3 TFLOP/s fp32 FMA spec / 2).  Only the cheap instruction class (add / sub /
# and / or / xor / not / lshr / ashr / mov / cndmask / bitop3 / f32 add-mul-fma) reaches it, and only in a stream of its own
# kind; everything else (shifts left, 3-input integer ops, min / max, mad24, compares, perm, alignbit, bcnt, every packed,
# DPP and SDWA form) takes 4 cycles (profiles/r03_issue_rate_ubench.log).
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9
N_SIMDS = 256 * 4
STREAM_FRAMES_C4 = 4096


def config_table():
    import image_processing_amd as ip
    from image_processing_amd import _lib as L
    from image_processing_amd.synth import DEFAULT_RANGE

    bands = [(20 * i, 20 * i + 19) for i in range(8)]
    return {
        "c1": dict(w=640, h=480, enc=L.ENC_BGR8, batch=256, in_flight=3, uniq=8, bpp=4, scaling="weak", hue_bands=((100, 130),),
                   pipe=dict(morph=ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE], cpu_budget_s=20.0,
                   metric="frames/sec at 640x480 BGR8, HSV inRange + 3x3 erode + 3x3 dilate + centroid (no blur)",
                   workload="configs[0] (the reference's CPU-runnable case, pose_estimate.cpp:203-206,405-415 with one erode and one "
                            "dilate): 640x480 BGR8, batches of 256 frames, BGR2HSV + inRange + erode3 + dilate3 + moments; the CPU leg "
                            "(`cpu_baseline`) is this configuration's subject, `value` is the GPU leg (device-resident) beside it"),
        "c2": dict(w=1920, h=1200, enc=L.ENC_BGR8, batch=256, uniq=8, bpp=4, scaling="weak", hue_bands=((100, 130),),
                   pipe=dict(post_blur_k=5, morph=ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE],
                   metric="frames/sec at 1920x1200 BGR8, full chain (5x5 Gaussian, HSV inRange, 3x3 erode+dilate, moments)",
                   workload="configs[1]: 1920x1200 BGR8, device-resident batch of 256 frames per GPU per step, post-HSV 5x5 "
                            "Gaussian + inRange + erode3 + dilate3 + moments, masks written to HBM"),
        "c3": dict(w=2448, h=2048, enc=L.ENC_BAYER_RGGB8, batch=64, uniq=4, bpp=2, scaling="weak", hue_bands=((100, 130),),
                   pipe=dict(post_blur_k=5, morph=ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE],
                   metric="frames/sec at 2448x2048 BayerRG8, demosaic + full chain (device-resident leg)",
                   workload="configs[2]: 2448x2048 BayerRG8 mosaics, device-resident batch of 64 frames per GPU per step, demosaic "
                            "inside the chain kernel + 5x5 Gaussian + inRange + erode3 + dilate3 + moments; the streamed leg "
                            "(pinned slots, H2D per frame) is reported under `streamed`"),
        "c4": dict(w=3840, h=2160, enc=L.ENC_BGR8, batch=64, uniq=8, bpp=4, scaling="strong", hue_bands=((100, 130),),
                   pipe=dict(post_blur_k=5, morph=ip.MORPH_BASELINE), ranges=[DEFAULT_RANGE], stream=STREAM_FRAMES_C4,
                   metric="frames/sec at 3840x2160 BGR8, 4096-frame stream sharded across the GPUs, full chain",
                   workload="configs[3]: one step = a 4096-frame 3840x2160 BGR8 stream (a pool of 64 device-resident frames, "
                            "cycled) cut into contiguous per-rank blocks by shard_bounds, device batches of 64 frames, two in "
                            "flight; per-frame centroids gathered on rank 0 by frame id after the timed region"),
        # not BASELINE configs: the reference nodes' own chains at the headline frame shape (two launches per batch)
        "pp": dict(w=1920, h=1200, enc=L.ENC_BGR8, batch=256, in_flight=3, uniq=8, bpp=4, scaling="weak", hue_bands=((100, 130),),
                   pipe=dict(pre_blur_k=3, post_blur_k=5, morph=ip.MORPH_POLYGON_PATH), ranges=[DEFAULT_RANGE],
                   metric="frames/sec at 1920x1200 BGR8, polygon_path chain (3x3 Gaussian, HSV, 5x5 Gaussian, inRange, 14 erode/dilate, moments)",
                   workload="objectPolygonPath.cpp:182-188,414-433 at 1920x1200 BGR8, device-resident batch of 256 frames per GPU per "
                            "step: front kernel (blur3, HSV, blur5, inRange -> bit planes) + bit-plane morphology (E3x4 D8x2 E3x2 D8x2 "
                            "E3x4 merged into 5 windows, moments, mask bytes)"),
        "pe": dict(w=1920, h=1200, enc=L.ENC_BGR8, batch=256, in_flight=3, uniq=8, bpp=4, scaling="weak", hue_bands=((100, 130),),
                   pipe=dict(post_blur_k=5, morph=ip.MORPH_POSE_ESTIMATE), ranges=[DEFAULT_RANGE],
                   metric="frames/sec at 1920x1200 BGR8, pose_estimate chain with its blur and morphOps enabled",
                   workload="pose_estimate.cpp:203-206,405-415 (commented lines enabled) at 1920x1200 BGR8, device-resident batch of 256 "
                            "frames per GPU per step: front kernel (HSV, blur5, inRange -> bit planes) + bit-plane morphology (E3x2 D8x2 "
                            "E3x2 merged into 3 windows, moments, mask bytes)"),
        "c5": dict(w=7680, h=4320, enc=L.ENC_BGR8, batch=4, uniq=1, bpp=11, scaling="weak", hue_bands=tuple(bands),
                   pipe=dict(post_blur_k=15, morph=[ip.E(9, L.ELLIPSE), ip.D(9, L.ELLIPSE)]),
                   ranges=[((lo, 100, 80), (hi, 255, 255)) for lo, hi in bands],
                   metric="frames/sec at 7680x4320 BGR8, stress chain (15x15 Gaussian, 8 ranges, 9x9 elliptical erode+dilate, moments)",
                   workload="configs[4] operators: 7680x4320 BGR8, device-resident batch of 4 frames per GPU per step, HSV + 15x15 "
                            "Gaussian + 8 inRange planes + 9x9 ellipse erode + dilate + 8 moment sets, 8 masks written"),
    }


# ---------------------------------------------------------------------------------------------------------------------
# parent: start one rank process per GPU (never initialises HIP itself)
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n: int, argv: list[str]) -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile

    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), BENCH_SPAWNED="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        bad = []
        while not bad and any(p.poll() is None for p in procs):
            time.sleep(0.05)
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad:  # a rank died: the others would wait for it in the next barrier until gloo's timeout -- end exactly those
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        out0.seek(0)
        sys.stdout.write(out0.read().decode())
        sys.stdout.flush()
    if bad:
        print(f"bench.py: rank processes failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline leg (rank 0, N = 1): the oracle, as the thing timed beside the GPU path -- never part of it
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(cfg: dict, frames, threads: int, budget_s: float = 12.0) -> dict:
    """Times the CPU port (oracle/cpu_ref.c built -O3) of the same chain on a bounded sample of the same frames."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import Oracle

    orc = Oracle(fast=True)
    pipe = cfg["pipe"]
    morph = [(m.kind, m.shape, m.kw, m.kh, -1, -1) for m in pipe["morph"]]
    enc = cfg["enc"]

    def one(i):
        return orc.chain(frames[i % len(frames)], encoding=enc, pre_blur_k=pipe.get("pre_blur_k", 0),
                         post_blur_k=pipe.get("post_blur_k", 0), morph=morph, ranges=cfg["ranges"], stages=False)["moments"]

    t0 = time.perf_counter()
    one(0)
    t1 = time.perf_counter() - t0  # one frame on one thread
    per_thread = max(1, min(8 if budget_s <= 12.0 else 4096, int(budget_s / max(t1, 1e-3))))
    n_frames = per_thread * threads
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL inside the C call
        list(ex.map(one, range(n_frames)))
    dt = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), model)
    except OSError:
        pass
    return {"value": n_frames / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n_frames} frames of {cfg['w']}x{cfg['h']}, same chain, oracle/cpu_ref.c -O3 x86-64-v3 (scalar restatement of "
                      f"OpenCV's arithmetic, not a SIMD OpenCV build), {threads} threads, one frame per thread at a time; "
                      "OpenCV itself is absent from the image",
            "value_1_thread": 1.0 / t1, "cpu_model": model}


def kernel_name(cfg: dict, plan: str) -> str:
    from image_processing_amd import _lib as L

    if plan.startswith("fused chain kernel"):
        pipe, nr = cfg["pipe"], len(cfg["ranges"])
        cap = nr if nr <= 2 else 4 if nr <= 4 else 8
        bayer = "true" if cfg["enc"] == L.ENC_BAYER_RGGB8 else "false"
        return (f"ipf::k_fused_chain<{3 if pipe.get('pre_blur_k', 0) == 3 else 1},{5 if pipe.get('post_blur_k', 0) == 5 else 1},"
                f"false,{cap},3,false,0,{bayer}>")
    return f"all chain kernels of one batch ({plan})"


def build_hashes() -> dict:
    """Identity of the kernel build this process runs: sha256[:16] of the loaded libimgproc_hip.so and of the sources it is
    built from (csrc/* + include/imgproc.h, names and contents).  scripts/pmc_collect.py records the same two."""
    import hashlib

    from image_processing_amd import _lib as L

    out = {"lib_sha16": None, "src_sha16": None}
    if os.path.exists(L.LIB_PATH):
        with open(L.LIB_PATH, "rb") as f:
            out["lib_sha16"] = hashlib.sha256(f.read()).hexdigest()[:16]
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "image_processing_amd", "csrc")
    files = sorted(os.path.join(csrc, n) for n in os.listdir(csrc)) + [os.path.join(ROOT, "include", "imgproc.h")]
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    out["src_sha16"] = h.hexdigest()[:16]
    return out


def offline_counters(name: str):
    """PMC figures of this configuration's chain kernel(s) from the committed rocprofv3 --pmc passes (profiles/): HBM
    bytes, VALU instructions and the clock per launch.  They are properties of the kernel build and the workload shape,
    not of this run, so they are quoted only when they were collected on THIS build: the summary's lib_sha16 must equal
    the loaded library's hash (or its src_sha16 the hash of the sources the library is built from); otherwise
    (None, reason)."""
    for name_ in ("r03_pmc_summary.json", "r02_pmc_summary.json"):
        path = os.path.join(ROOT, "profiles", name_)
        if os.path.exists(path):
            break
    else:
        return None, "no profiles/r0x_pmc_summary.json"
    with open(path) as f:
        pmc = json.load(f).get(name)
    rel = os.path.relpath(path, ROOT)
    if not pmc:
        return None, f"{rel} does not cover this configuration"
    mine = build_hashes()
    same_lib = pmc.get("lib_sha16") and pmc.get("lib_sha16") == mine["lib_sha16"]
    same_src = pmc.get("src_sha16") and pmc.get("src_sha16") == mine["src_sha16"]
    if not (same_lib or same_src):
        return None, (f"{rel} was collected on another kernel build (lib {pmc.get('lib_sha16')}, src {pmc.get('src_sha16')}; loaded: "
                      f"lib {mine['lib_sha16']}, src {mine['src_sha16']}): not quoted")
    pmc = dict(pmc, matched_on="lib_sha16" if same_lib else "src_sha16")
    return pmc, rel


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args) -> None:
    import numpy as np

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dry = os.environ.get("BENCH_DRY_RUN") == "1"  # CPU rehearsal of the launch / rendezvous / relay logic: no GPU work

    # before anything touches the GPU: this rank's threads (the library's staging-copy helpers included) and every
    # page-locked buffer allocated below stay on the NUMA node its GPU hangs off
    from image_processing_amd.shard import pin_to_gpu_numa

    gpu_index = 0 if os.environ.get("BENCH_ALL_RANKS_ON_GPU0") == "1" else local_rank
    numa = pin_to_gpu_numa(gpu_index, sysfs=os.environ.get("BENCH_SYSFS_ROOT", "/sys")) if not args.no_numa else {"pinned": False, "skipped": True}

    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections ("[Gloo] Rank 0 is connected to ...") on stdout, which has to carry exactly one
        # JSON line: while the group forms, file descriptor 1 points at stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    import image_processing_amd as ip
    from image_processing_amd import _lib as L
    from image_processing_amd.shard import centroids, gather_by_frame, shard_bounds
    from image_processing_amd.synth import synth_bayer_rggb, synth_frame

    cfg = config_table()[args.config]
    if args.batch:  # experiments only: the configurations' own batch sizes are the ones profiles/ and DESIGN.md quote
        cfg = dict(cfg, batch=args.batch, workload=cfg["workload"] + f" [--batch {args.batch} overrides the batch size]")
    W, H, enc, pool_n, uniq = cfg["w"], cfg["h"], cfg["enc"], cfg["batch"], cfg["uniq"]
    cn = 1 if enc == L.ENC_BAYER_RGGB8 else 3
    nr = len(cfg["ranges"])
    algo_bytes = W * H * cfg["bpp"]
    strong = cfg["scaling"] == "strong"

    # the batches of one step: (offset into the device frame pool, frames, first stream frame id)
    if strong:
        a, b = shard_bounds(cfg["stream"], rank, world)
        batches, cur = [], a
        while cur < b:
            n = min(pool_n - cur % pool_n, b - cur)
            batches.append((cur % pool_n, n, cur))
            cur += n
        frames_per_step_total = cfg["stream"]
    else:
        a, b = rank * pool_n, (rank + 1) * pool_n
        batches = [(0, pool_n, a)]
        frames_per_step_total = pool_n * world
    n_local = b - a

    if dry:
        dt = 1e-3 * (rank + 1) * args.steps
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        local = np.zeros((n_local, nr, 3), np.int64)
        local[:, :, 0] = 1
        local[:, :, 1] = np.arange(a, b)[:, None]
        gathered = gather_by_frame(local, frames_per_step_total, rank, world, dist)
        if rank == 0:
            assert (gathered[:, 0, 1] == np.arange(frames_per_step_total)).all()
            print(json.dumps({"metric": cfg["metric"], "value": frames_per_step_total * args.steps / dt, "unit": "frames/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                              "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "u8",
                              "data": "dry run: no GPU work", "config": {"workload": cfg["workload"], "dry_run": True},
                              "numa": dict(numa, affinity_now=sorted(os.sched_getaffinity(0))),
                              **({"streamed": {"value": frames_per_step_total / dt, "unit": "frames/s", "pcie_gbps": 0.0, "slots": 3,
                                               "frames": frames_per_step_total, "dry_run": True}} if args.streamed else {})}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    if os.environ.get("BENCH_ALL_RANKS_ON_GPU0") == "1":  # rehearsal of the N>1 path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # synthetic frames: `uniq` seeded frames per rank, replicated (rolled) to a pool of distinct HBM buffers
    seed = 0x1A6E5EED + 1000 * rank
    if cn == 1:
        base_np = np.stack([synth_bayer_rggb(W, H, i, seed=seed, hue_bands=cfg["hue_bands"]) for i in range(uniq)])
    else:
        base_np = np.stack([synth_frame(W, H, i, seed=seed, hue_bands=cfg["hue_bands"]) for i in range(uniq)])
    base = torch.from_numpy(base_np).to(dev)
    frames = torch.empty((pool_n,) + tuple(base.shape[1:]), dtype=torch.uint8, device=dev)
    for i in range(pool_n):
        frames[i] = torch.roll(base[i % uniq], (2 if cn == 1 else 1) * (i // uniq), dims=0)  # even shifts keep the mosaic phase
    masks = torch.empty((pool_n, nr, H, W), dtype=torch.uint8, device=dev)
    # with ticket 1 on a stream of its own the two batches in flight write at the same time: one mask buffer each
    # device batches in flight (tickets of the C ABI): 2, or 3 where measured better (the two-launch chains and the small frames
    # of c1: profiles/r03_in_flight_sweep.log); 4 is slower everywhere
    DEPTH = max(1, min(L.MAX_TICKETS, int(os.environ.get("BENCH_IN_FLIGHT", cfg.get("in_flight", 2)))))
    mask_bufs = [masks] + [torch.empty_like(masks) for _ in range(DEPTH - 1)]
    torch.cuda.synchronize()
    frame_bytes, row_bytes = H * W * cn, W * cn

    g = ip.ImgProc(W, H, device=local_rank, n_slots=1, max_batch=pool_n)
    g.set_pipeline(encoding=enc, force_unfused=args.unfused, **cfg["pipe"])
    g.set_ranges(cfg["ranges"])
    g.set_profiling(True)
    plan = g.plan

    # the C ABI fills these host arrays with the per-frame moments / centroids; two, because two batches are in flight:
    # batch k+1 is put on the stream before batch k's results are waited for, so the GPU never idles between them
    raws = [(L.Result * pool_n)() for _ in range(DEPTH)]

    def enqueue(k):
        off, n, _ = batches[k % len(batches)]
        g.batch_enqueue(k % DEPTH, frames.data_ptr() + off * frame_bytes, frame_bytes, row_bytes, W, H, n, mask_bufs[k % DEPTH].data_ptr())

    def wait(k):
        return g.batch_wait(k % DEPTH, raw=raws[k % DEPTH])

    def fence():
        g.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # The GPU's clocks take tens of milliseconds of sustained load to settle (measured: the same kernel runs 0.87 ms in
    # a 23-step run and 0.78 ms from the ~40th step on).  The metric is steady-state throughput, so the device is kept
    # busy with the same work for PREROLL_S before the W warmup steps; nothing of it is timed or reported.
    t_pre = time.perf_counter()
    k_pre = 0
    while time.perf_counter() - t_pre < PREROLL_S or k_pre < DEPTH:  # the same in-flight pattern as the timed loop
        enqueue(k_pre)
        if k_pre >= DEPTH - 1:
            wait(k_pre - DEPTH + 1)
        k_pre += 1
    for k in range(max(0, k_pre - DEPTH + 1), k_pre):
        wait(k)
    nb = len(batches)
    n_warm = args.warmup * nb
    for k in range(n_warm):
        enqueue(k)
        if k >= DEPTH - 1:
            wait(k - DEPTH + 1)
    for k in range(max(0, n_warm - DEPTH + 1), n_warm):
        wait(k)
    fence()
    kernel_ms, kernel_frames, kernel_spans = [], [], []
    g.set_profiling(True)  # time zero of the kernel spans: float milliseconds keep their resolution over a short region
    local = np.zeros((n_local, nr, 3), np.int64)
    total = args.steps * nb

    def collect(k):
        res = wait(k)
        kernel_ms.append(g.last_kernel_ms())  # HIP events recorded on the batch's stream around its chain kernel(s)
        kernel_spans.append(g.last_kernel_span())  # the same two events on the device clock
        off, n, first = batches[k % nb]
        kernel_frames.append(n)
        if k >= total - nb:  # the last step's results, by stream frame id
            for i in range(n):
                for r in range(nr):
                    m = res[i].r[r]
                    local[first - a + i, r] = (m.m00, m.m10, m.m01)

    t0 = time.perf_counter()
    for k in range(total):  # every batch's results reach the host inside the timed region
        enqueue(k)
        if k >= DEPTH - 1:
            collect(k - DEPTH + 1)
    for k in range(max(0, total - DEPTH + 1), total):
        collect(k)
    g.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()

    # host-side gather of the last step's per-frame moments, ordered by global frame id.  Outside the timed region: the
    # data path itself has no inter-rank step.
    t_g = time.perf_counter()
    gathered = gather_by_frame(local, frames_per_step_total, rank, world, dist)
    gather_ms = (time.perf_counter() - t_g) * 1e3
    n_found = int((centroids(gathered)[:, 0, 0] >= 0).sum()) if rank == 0 else 0

    streamed = None
    if args.streamed and cn == 3:
        streamed = streamed_rank_leg(cfg, frames, batches, local, a, local_rank, world, dist, nb)
    checked = False
    if args.check and rank == 0:
        # one frame of the last batch against the oracle, after the timed region (every run, unless --no-check): the
        # number above is worth nothing if the masks and moments it produced are wrong
        from oracle import Oracle

        i = min(9, batches[-1][1] - 1)
        f = frames[batches[-1][0] + i].cpu().numpy()
        pipe = cfg["pipe"]
        morph = [(m.kind, m.shape, m.kw, m.kh, -1, -1) for m in pipe["morph"]]
        exp = Oracle(fast=True).chain(f, encoding=enc, pre_blur_k=pipe.get("pre_blur_k", 0), post_blur_k=pipe.get("post_blur_k", 0),
                                      morph=morph, ranges=cfg["ranges"], stages=False)
        got = mask_bufs[(total - 1) % DEPTH][i].cpu().numpy()
        for r in range(nr):
            assert (got[r] == exp["mask"][r]).all(), f"bench mask of range {r} != oracle"
            idx = batches[-1][2] - a + i
            assert [int(v) for v in local[idx, r]] == [int(v) for v in exp["moments"][r][:3]], "bench moments != oracle"
        checked = True

    # The classical per-launch figure beside the overlapped one: the first batch a few times, one at a time (nothing else on
    # the GPU), HIP events around its kernels.  After the timed region and the check: `value` does not see it.
    alone_ms = []
    t_pre = time.perf_counter()
    while args.alone and (time.perf_counter() - t_pre < PREROLL_S or len(alone_ms) < 12):  # the clocks have dropped during the check: same pre-roll
        enqueue(0)
        wait(0)
        alone_ms.append(g.last_kernel_ms()[0])
    alone_ms = alone_ms[-12:]
    alone_frames = batches[0][1]

    if rank == 0:
        fps = frames_per_step_total * args.steps / dt
        ms_sum = float(sum(m for m, _ in kernel_ms))
        k_ms = ms_sum / len(kernel_ms)  # per batch launch (one chain kernel on the fused path)
        launches = kernel_ms[-1][1]
        bytes_per_launch = algo_bytes * (sum(kernel_frames) / len(kernel_frames))
        # The two tickets run on two streams, so the kernels of consecutive batches overlap on the GPU: the time the chain
        # kernels took is the UNION of their spans, not the sum of their lengths (equal when nothing overlaps).
        busy_ms, end = 0.0, -1.0
        for t0_, t1_ in sorted(kernel_spans):
            if t1_ > end:
                busy_ms += t1_ - max(t0_, end)
                end = t1_
        in_flight = ms_sum / busy_ms
        k_eff = busy_ms / len(kernel_spans)  # kernel time per batch while the chain kernels run
        achieved = algo_bytes * sum(kernel_frames) / (busy_ms * 1e-3) / 1e9
        fused = plan.startswith("fused chain kernel")
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None, "traffic_source": None,
                "kernel": kernel_name(cfg, plan), "kernel_ms_per_launch": k_ms if fused else None,
                "kernel_ms_per_batch": k_ms, "launches_per_batch": launches,
                "batches_in_flight": in_flight, "kernel_ms_per_batch_effective": k_eff,
                "one_batch_at_a_time": None if not alone_ms else {"kernel_ms_per_batch": sum(alone_ms) / len(alone_ms),
                                        "achieved": algo_bytes * alone_frames / (sum(alone_ms) / len(alone_ms) * 1e-3) / 1e9,
                                        "frac": algo_bytes * alone_frames / (sum(alone_ms) / len(alone_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                        "note": "the same batch with nothing else in flight (the last 12 of a 0.15 s run of single batches after the timed region, HIP events "
                                                "around the kernels of each): bytes / launch duration in the plain sense; alone the chain "
                                                "kernel is cut into ~100-row segments, overlapped into ~300-row ones"},
                "timing": "HIP events around each batch's kernels on the batch's own stream (kernel_ms_per_batch = average span, what "
                          "rocprofv3 --kernel-trace --stats reports per launch); consecutive batches run on two streams and overlap on "
                          "the GPU, so achieved = algorithmic bytes of all batches / union of the spans "
                          "(kernel_ms_per_batch_effective = union / batches = kernel_ms_per_batch / batches_in_flight)",
                "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_bytes_per_px": cfg["bpp"]}
        pmc, src = offline_counters(args.config if not args.unfused else "none")
        if pmc is None:
            roof["traffic_source"] = src  # why nothing is quoted
        elif abs(pmc.get("frames_per_launch", 0) - bytes_per_launch / algo_bytes) >= 0.5:
            roof["traffic_source"] = f"{src} covers {pmc.get('frames_per_launch')} frames per launch, this run {bytes_per_launch / algo_bytes:.0f}: not quoted"
        else:
            # HBM bytes from rocprofv3 --pmc passes of this same command (FETCH_SIZE / WRITE_SIZE, separate runs, the
            # guide's gfx950 corrections); VALU instructions and the clock (GRBM_GUI_ACTIVE) likewise -- offline figures
            # of this kernel build (hash-checked), not of this run
            roof["traffic"] = pmc.get("hbm_bytes_per_launch")
            roof["traffic_source"] = (f"{src} (offline rocprofv3 --pmc passes of this command on this kernel build, matched on "
                                      f"{pmc['matched_on']} {pmc.get(pmc['matched_on'])})")
            if pmc.get("valu_insts_per_launch"):
                n_valu = pmc["valu_insts_per_launch"]
                lane_ops = n_valu * 64 / (k_eff * 1e-3)
                clock = pmc.get("clock_ghz") or 2.4
                valu = {"achieved_lane_ops_per_s": lane_ops, "peak": VALU_PEAK_LANE_OPS, "frac": lane_ops / VALU_PEAK_LANE_OPS,
                        "peak_note": "machine peak: 256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz (one wave64 instruction per 2 cycles); "
                                     "reached only by a stream of cheap-class instructions",
                        "valu_insts_per_launch": n_valu,
                        "lane_ops_per_px": n_valu * 64 / (bytes_per_launch / cfg["bpp"]),
                        "clock_ghz": clock, "clock_source": "GRBM_GUI_ACTIVE / 8 / kernel time (offline pass)" if pmc.get("clock_ghz") else "assumed",
                        "cycles_per_valu_inst": k_eff * 1e-3 * clock * 1e9 * N_SIMDS / n_valu,
                        "insts_source": roof["traffic_source"]}
                cheap = pmc.get("cheap_class_share")  # static share of cheap-class instructions in the main loop (ISA count)
                if cheap is not None:
                    # issue-class floor: every cheap-class instruction at 2 cycles, every other at 4, nothing else in the way
                    floor_ms = n_valu * (2.0 * cheap + 4.0 * (1.0 - cheap)) / N_SIMDS / (clock * 1e9) * 1e3
                    valu["issue_class"] = {"cheap_class_share": cheap, "floor_ms_cheap2_full4": floor_ms, "frac_of_floor": floor_ms / k_eff,
                                           "note": "floor = every cheap-class instruction at 2 cycles, every other at 4 (frac_of_floor = floor / "
                                                   "measured kernel time).  A mixed stream does not reach it (profiles/r03_issue_rate_ubench.log: "
                                                   "3.6-4.1 cycles per instruction for every cheap / full-rate mix in runs of 1 to 48), and three "
                                                   "bit-identical builds of this kernel with 422-462 M instructions and 20-44 % cheap share run "
                                                   "within 1 % of each other (DESIGN.md 3.1); without any HBM traffic the kernel keeps 96 % of its time"}
                roof["valu"] = valu
        out = {
            "metric": cfg["metric"], "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": cfg["workload"] + f"; {DEPTH} batches in flight, each on its own stream (results of batch k are "
                                                    f"collected after batch k+{DEPTH - 1} is enqueued)",
                       "batches_in_flight": DEPTH,
                       "name": args.config, "checked": checked, "frames_per_step": frames_per_step_total, "frames_per_batch": pool_n, "width": W,
                       "height": H, "n_ranges": nr, "untimed_preroll_s": PREROLL_S,
                       "path": plan if not args.unfused else f"stage-by-stage ({launches} launches per batch)",
                       "sharding": f"frames x{world} ranks ({cfg['scaling']}), no collective; {frames_per_step_total} centroids "
                                   f"gathered on rank 0 after the timed region in {gather_ms:.2f} ms ({n_found} with an object)"},
            "roofline": roof,
            "achieved_gbps_whole_job": fps * algo_bytes / 1e9,
        }
        out["numa"] = numa
        if streamed is not None:
            out["streamed"] = streamed
        if args.config == "c3" and world == 1 and not args.no_streamed:
            out["streamed"] = streamed_leg(cfg, base_np, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(cfg, base_np, threads, budget_s=cfg.get("cpu_budget_s", 12.0))
        print(json.dumps(out), flush=True)
    g.close()
    if dist is not None:
        dist.destroy_process_group()


def streamed_rank_leg(cfg: dict, frames_dev, batches, local, first_frame: int, device: int, world: int, dist, nb: int, slots: int = 3) -> dict:
    """The streamed mode of a sharded configuration (SURVEY §8d "two reported modes", §8e): every rank feeds ITS block of
    the step's frames from page-locked host memory through imgproc_submit_pinned (`slots` slots, each on its own stream:
    upload of frame k+1 overlaps the chain kernel of frame k), moments back per frame.  PCIe-bound; timed like the main
    leg (barrier, max over ranks) and reported beside the device-resident `value`, never as it.  The per-frame m00 / m10 /
    m01 must equal the device-resident leg's for the same stream frame ids."""
    import ctypes as C

    import numpy as np
    import torch

    import image_processing_amd as ip
    from image_processing_amd import _lib as L

    W, H = cfg["w"], cfg["h"]
    pool_n = frames_dev.shape[0]
    n_local = sum(n for _, n, _ in batches[:nb])
    with ip.ImgProc(W, H, device=device, n_slots=slots) as g:
        g.set_pipeline(encoding=cfg["enc"], **cfg["pipe"])
        g.set_ranges(cfg["ranges"])
        pin = g.host_alloc((pool_n, H, W, 3))  # the same pool of frames, page-locked, on this rank's NUMA node
        pin[:] = frames_dev.cpu().numpy()
        addr0, fstride, rstride = pin.ctypes.data, pin.strides[0], pin.strides[1]
        order = [(off + i) for off, n, _ in batches[:nb] for i in range(n)]  # pool index of every frame of my block, in stream order
        res = (L.Result * max(n_local, 1))()
        lib, ctx = g.lib, g._ctx

        def run_all(count):
            # submit / wait pipeline over `count` frames: wait for slot s just before it is reused, then drain
            waited = 0
            for i in range(count):
                if i >= slots:
                    rc = lib.imgproc_wait(ctx, i % slots, C.byref(res[i - slots]))
                    assert rc == 0, lib.imgproc_last_error(ctx)
                    waited += 1
                rc = lib.imgproc_submit_pinned(ctx, i % slots, addr0 + order[i] * fstride, rstride, W, H)
                assert rc == 0, lib.imgproc_last_error(ctx)
            for i in range(waited, count):
                rc = lib.imgproc_wait(ctx, i % slots, C.byref(res[i]))
                assert rc == 0, lib.imgproc_last_error(ctx)

        run_all(min(n_local, 32))  # warm-up
        g.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        run_all(n_local)
        g.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # same frames, same chain: the streamed moments are the device-resident leg's (last step), frame by frame
        nr = len(cfg["ranges"])
        for i in range(n_local):
            for r in range(nr):
                m = res[i].r[r]
                assert (m.m00, m.m10, m.m01) == tuple(int(v) for v in local[i, r]), f"streamed frame {first_frame + i} != device-resident leg"
        del res
        g.host_free(pin)
    total = cfg.get("stream", pool_n * world) if cfg["scaling"] == "strong" else n_local * world
    fb = W * H * 3
    return {"value": total / dt, "unit": "frames/s", "pcie_gbps": total * fb / dt / 1e9, "pcie_gbps_per_gpu": total * fb / dt / 1e9 / world,
            "slots": slots, "frames": total, "frames_per_rank": n_local, "seconds": dt,
            "note": "every rank streams its block of the step's frames from page-locked host memory through imgproc_submit_pinned "
                    "(3 slots, H2D of frame k+1 overlaps the chain kernel of frame k), per-frame moments back; barrier + max over "
                    "ranks; per-frame moments equal the device-resident leg's (asserted); PCIe Gen5 x16 practical peak ~55 GB/s per GPU"}


def streamed_leg(cfg: dict, base_np, device: int, n: int = 96, slots: int = 3) -> dict:
    """configs[2]'s upload leg: host frames -> the slots' pinned staging buffers -> hipMemcpyAsync on each slot's stream,
    overlapped with the chain kernel of the slot before -> 48 B of moments back (imgproc_process_batch).  PCIe-bound;
    reported beside the device-resident number, never as `value`."""
    import numpy as np

    import image_processing_amd as ip
    from image_processing_amd import _lib as L

    W, H = cfg["w"], cfg["h"]
    host = np.ascontiguousarray(np.concatenate([base_np] * (n // len(base_np))))
    with ip.ImgProc(W, H, device=device, n_slots=slots) as g:
        g.set_pipeline(encoding=cfg["enc"], **cfg["pipe"])
        g.set_ranges(cfg["ranges"])
        res = (L.Result * len(host))()

        def run():
            rc = g.lib.imgproc_process_batch(g._ctx, host.ctypes.data, host.strides[0], host.strides[1], W, H, len(host), res)
            assert rc == 0, g.lib.imgproc_last_error(g._ctx)

        run()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        dt = (time.perf_counter() - t0) / reps
        # the same stream from page-locked frames (a camera driver's DMA ring): imgproc_submit_pinned, no staging memcpy
        import ctypes as C

        pin = g.host_alloc(host.shape)
        pin[:] = host
        res2 = (L.Result * len(host))()

        def run_pinned():
            for i in range(len(host) + slots):
                if i >= slots:
                    rc = g.lib.imgproc_wait(g._ctx, (i - slots) % slots, C.byref(res2[i - slots]))
                    assert rc == 0, g.lib.imgproc_last_error(g._ctx)
                if i < len(host):
                    rc = g.lib.imgproc_submit_pinned(g._ctx, i % slots, pin[i].ctypes.data, pin[i].strides[0], W, H)
                    assert rc == 0, g.lib.imgproc_last_error(g._ctx)

        run_pinned()
        t0 = time.perf_counter()
        for _ in range(reps):
            run_pinned()
        dtp = (time.perf_counter() - t0) / reps
        assert all(res2[i].r[0].m00 == res[i].r[0].m00 for i in range(len(host)))
        g.host_free(pin)
    fps, fps_p = len(host) / dt, len(host) / dtp
    return {"value": fps, "unit": "frames/s", "pcie_gbps": fps * host[0].nbytes / 1e9, "slots": slots, "frames": len(host),
            "pinned_value": fps_p, "pinned_pcie_gbps": fps_p * host[0].nbytes / 1e9,
            "note": "value: imgproc_submit (host frames -> pinned staging, memcpy by the caller + 3 helper threads -> H2D per slot stream -> "
                    "chain kernel -> moments); pinned_value: imgproc_submit_pinned (the caller's frames are page-locked, no staging "
                    "memcpy); PCIe Gen5 x16 practical peak ~55 GB/s"}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5", "pp", "pe"], default="c2")
    ap.add_argument("--batch", type=int, default=0, help="override the configuration's frames per batch (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streamed", action="store_true", help="c3: skip the streamed (PCIe) leg")
    ap.add_argument("--unfused", action="store_true", help="time the stage-by-stage path instead")
    ap.add_argument("--streamed", action="store_true",
                    help="BGR configurations (c4: the sharded stream): add the streamed leg -- every rank feeds its frames from "
                         "page-locked host memory through imgproc_submit_pinned; reported under `streamed`, never as `value`")
    ap.add_argument("--no-numa", action="store_true", help="leave the CPU affinity alone (default: pin each rank to its GPU's NUMA node)")
    ap.add_argument("--no-alone", dest="alone", action="store_false",
                    help="skip the one-batch-at-a-time leg after the timed region (the profiling scripts: its launches would mix into the per-kernel averages)")
    ap.add_argument("--no-check", dest="check", action="store_false",
                    help="skip the oracle check of one frame of the last batch (it runs after the timed region by default)")
    ap.add_argument("--check", dest="check", action="store_true", help="(default) verify one frame of the last batch against the oracle")
    ap.set_defaults(check=True)
    args = ap.parse_args()
    defaults = {"c1": (200, 40), "c2": (100, 20), "c3": (50, 10), "c4": (5, 1), "c5": (20, 4), "pp": (60, 10), "pe": (60, 10)}[args.config]
    if args.steps is None:
        args.steps = defaults[0]
    if args.warmup is None:
        args.warmup = defaults[1]
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start the ranks from here.  This process has not imported torch or touched HIP and
        # never will; the ranks are fresh child processes, not re-execs.
        argv = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--config", args.config]
        argv += ["--no-cpu-baseline"] if args.no_cpu_baseline else []
        argv += ["--unfused"] if args.unfused else []
        argv += ["--streamed"] if args.streamed else []
        argv += ["--no-numa"] if args.no_numa else []
        argv += ["--no-streamed"] if args.no_streamed else []
        argv += [] if args.check else ["--no-check"]
        return spawn_ranks(args.gpus, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: using the launcher's world size", file=sys.stderr)
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
