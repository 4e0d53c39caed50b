"""Headline benchmark: ray-samples/s of the HIP render path on BASELINE config 3
(800x800 view, 64 coarse + 128 importance samples, 8x256 coarse and fine MLPs, synthetic weights).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A step renders N frames (N = number of GPUs; one frame at N = 1): every frame is sharded by row tile over the
N ranks (rank r renders rows [r*800/N, (r+1)*800/N) of all N poses in ONE launch) and one RCCL gather brings
the tiles to rank 0, so per-GPU work is fixed (640 000 rays = 122.88 M ray-samples per step): weak scaling.
Inputs (weights, tables) are resident in HBM before the timed region; the only host->device traffic in a step
is the 64-byte poses.  Rank 0 prints ONE JSON line.

For N > 1 the line also carries, measured AFTER the timed region of the headline number:
`strong`  = BASELINE config 4: ONE 800x800 frame sharded N ways (80 000 rays per rank at N = 8) + gather, ms per frame
            (max over ranks) and the per-rank kernel time, i.e. the latency case where the last round of workgroups
            dominates - weak scaling cannot show a loss there by construction;
`in_process` = the same single frame through nwe_render_tiled from rank 0 alone (N contexts in one process,
            hipMemcpyPeerAsync into the frame): the path a GUI thread uses.  Best effort: an error is reported, not raised.

At N = 1 the line also carries `configs`: the other BASELINE.json configurations the driver's one GPU can run, measured
after the headline (kernel ms from the library's HIP events, best of three launches): C1 (64x64, 32 samples, 4x128, coarse
only), C2 (400x400, 64 coarse samples, 8x256), the YAML's 320x240 GUI frame (64+128) and C5 on one GPU (32 poses x 800x800
in ONE launch, frames/s).  They are parity-test cases (tests/test_gpu_parity.py), not bench lines; the numbers are there so
that the driver's record holds them.

`roofline` prices the render kernel (the only kernel of the path) against the dense fp16 MFMA peak with the
ALGORITHMIC FLOPs of the reference formulation (2 x GEMM MACs, SURVEY.md §8d: 1 186 816 FLOP per MLP
evaluation, 64 coarse + 192 fine evaluations per ray); its duration comes from HIP events recorded by the
library on the launch stream around each launch.  `cpu_baseline` times the oracle (torch CPU, the reference's
arithmetic) as BASELINE.md section 3 plans it: four reference chunks (32 768 rays) of the same frame, extrapolated linearly,
and C1 in full, on every host core this process may use (count stated).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H = W = 800
NS, NI = 64, 128
STRONG_STEPS = 5
LIBRARY_GEMM_F16_TFLOPS = 1330.0   # measured on this pool, tools/gemm_reference.py
POWER_CAPPED_F16_TFLOPS = 1650.0   # measured on this pool, see profiles/r01_ubench_mfma_power.txt
PEAK_F16_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
TRAFFIC_PROFILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc_traffic.json")
CPU_CHUNK = 8192           # inference.chunk = 1024*8 (nerf/configs/office_tokyo_config.yaml:41)
CPU_SAMPLE_RAYS = 4 * CPU_CHUNK   # >= 4 chunks of C3, extrapolated linearly (cost per ray is constant)


def sweep_pose(k: int, n: int) -> np.ndarray:
    """office_tokyo centre click, camera turned left by k*360/n degrees (GUI turn-left sweep)."""
    from nwe_amd import COORD, get_camera_poses_from_list_of_coordinates
    init = COORD(x=0.0, y=-0.5, z=-0.75 / np.cos(-10.0 / 180.0 * np.pi), yaw=0.0, pitch=-90.0, roll=0.0)
    hor = 30.0 + 360.0 * k / max(n, 1)
    return get_camera_poses_from_list_of_coordinates(init, [COORD(yaw=-hor)])[0].numpy()


def kernel_source_sha() -> str:
    """sha256 over the kernel sources: ties a committed counter profile to the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "nerf-workspaces-explorer_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def profiled_traffic():
    """(bytes per C3 launch, note): HBM-side bytes from separate rocprofv3 --pmc passes over this same command
    (tools/r03_profile.sh -> profiles/r03_pmc_traffic.json), used only when that profile was taken on the kernel sources
    in the tree; otherwise (None, why) - counters cannot be read inside this run."""
    try:
        prof = json.load(open(TRAFFIC_PROFILE))
    except (OSError, ValueError):
        return None, "no counter profile committed for this build"
    if prof.get("kernel_source_sha") != kernel_source_sha():
        return None, f"{os.path.basename(TRAFFIC_PROFILE)} was taken on kernel sources {prof.get('kernel_source_sha')}, not the ones in the tree"
    return float(prof["hbm_bytes_per_launch"]), prof.get("note", "")


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads() -> int:
    """The host cores this process may actually use: its affinity mask, capped by the cgroup CPU quota (the GPU box shows
    all 256 logical CPUs to a one-GPU job but schedules it on a 16-core quota: more threads than that only get throttled)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def progress(msg: str) -> None:
    """A line on stderr per phase: the JSON line is the only thing on stdout, and a run that is silent for minutes looks hung."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(sd_c, sd_f, pose):
    """the oracle on four 8192-ray reference chunks from the middle of the frame (the reference's
    chunking: 8192 rays per chunk, 32768 points per MLP call), and the C1 frame in full; all host threads."""
    from oracle import nerf_oracle as O
    threads = host_threads()
    torch.set_num_threads(threads)
    fx, fy, cx, cy = O.intrinsics(H, W)
    rays = O.create_rays(torch.from_numpy(pose)[None], H, W, fx, fy, cx, cy, 0.1, 10.0)[0]
    start = (H // 2) * W
    rays = rays[start:start + CPU_SAMPLE_RAYS].contiguous()
    t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
    cfg = O.RenderConfig(n_samples=NS, n_importance=NI, chunk=CPU_CHUNK)
    progress(f"cpu baseline: {CPU_SAMPLE_RAYS} rays of the frame through the oracle on {threads} threads")
    t0 = time.perf_counter()
    ref = O.render_rays(rays, t(sd_c), t(sd_f), cfg, keep=("rgb_fine", "raw_fine", "z_fine"))
    dt = time.perf_counter() - t0
    # C1 in full: 64x64, 32 coarse samples, coarse-only 4x128 (BASELINE.json configs[0], the reference's own CPU-runnable case)
    import nwe_amd
    c1 = t(nwe_amd.synthetic.make_state_dict(1000, 4, 128))
    f1 = O.intrinsics(64, 64)
    t1 = time.perf_counter()
    rays1 = O.create_rays(torch.from_numpy(pose)[None], 64, 64, *f1, 0.1, 10.0)[0]
    O.render_rays(rays1, c1, None, O.RenderConfig(n_samples=32, n_importance=0, chunk=CPU_CHUNK), keep=("rgb_coarse",))
    dt1 = time.perf_counter() - t1
    return rays, ref, dt, start, threads, dt1


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f16x1", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (C1, C2, GUI frame, C5 on one GPU)")
    ap.add_argument("--in-process-probe", type=int, default=0, metavar="N",
                    help="internal: render ONE frame through nwe_render_tiled on devices 0..N-1 from this process, print a JSON object, exit")
    ap.add_argument("--unfolded", action="store_true",
                    help="comparison only: evaluate _feature_linear as its own layer instead of folding it into the view layer at pack time")
    args = ap.parse_args()

    import torch.distributed as dist

    import nwe_amd
    from nwe_amd.dist import TileShardedRenderer, gather_tiles

    if args.in_process_probe:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sd_c = nwe_amd.synthetic.make_state_dict(1000, 8, 256)
        sd_f = nwe_amd.synthetic.make_state_dict(1001, 8, 256)
        print(json.dumps(in_process_frame(args, sd_c, sd_f, sweep_pose(0, 1), args.in_process_probe, torch.cuda.device_count())), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; must be set before the first HIP call
    # NWE_BENCH_BACKEND=gloo is a REHEARSAL switch for boxes with fewer GPUs than ranks (tests/test_gpu_dist.py runs two ranks
    # on the one GPU of the test box, which RCCL does not allow): ranks share devices, the gather goes through the host.
    backend = os.environ.get("NWE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    host_group = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        host_group = dist.new_group(backend="gloo")      # a barrier that parks no kernel on the GPUs (see the end of main)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    sd_c = nwe_amd.synthetic.make_state_dict(1000, 8, 256)
    sd_f = nwe_amd.synthetic.make_state_dict(1001, 8, 256)
    h = nwe_amd.NeRFReplicaInferenceHandler("office_tokyo", "synthetic", device=local_rank, precision=args.precision)
    h.set_sampling(NS, NI)
    if args.unfolded:
        h.debug_set_fold(False)
    h.initialize_models(state_dicts=(sd_c, sd_f))
    tsr = TileShardedRenderer(lambda poses, hh, ww, rows: h.render_batch(poses, hh, ww, rows=rows), rank, world)

    frames_per_step = world
    poses = np.stack([sweep_pose(k, frames_per_step) for k in range(frames_per_step)])
    kernel_ms = []

    parts = []                                            # per step: [(ms, rays)] of the launches the frame took

    def step():
        out = tsr.render_frames(poses, H, W)
        kernel_ms.append(h.renderer.last_kernel_ms())     # blocks until the launch has finished (HIP events on its stream)
        if args.precision != "f32":
            parts.append(h.renderer.last_launch_parts())
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        progress(f"{args.warmup} warm-up + {args.steps} timed steps of {frames_per_step} frame(s)")
    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    parts.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    rays_per_step = frames_per_step * H * W
    samples_per_step = rays_per_step * (NS + NI)
    evals_per_ray = NS + (NS + NI)
    value = samples_per_step * args.steps / elapsed
    k_ms = float(np.mean(kernel_ms))

    # ---- after the headline measurement: the single-frame (strong scaling) case, N > 1 only ---------------------------------
    strong = None
    rccl = None
    if world > 1:
        one = poses[:1]
        skm, gms = [], []
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            tsr.render_frames(one, H, W)
        fence()
        t1 = time.perf_counter()
        for _ in range(STRONG_STEPS):
            tile = tsr.render_local(one, H, W)
            skm.append(h.renderer.last_kernel_ms())
            ev0.record()                                  # torch's current stream: the one the gather is queued on
            gather_tiles(tile, H, rank, world)
            ev1.record()
            ev1.synchronize()
            gms.append(ev0.elapsed_time(ev1))
        fence()
        st = torch.tensor([time.perf_counter() - t1, float(np.mean(skm)), float(np.mean(gms))], dtype=torch.float64, device=red_dev)
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
        ms_frame = float(st[0].item()) / STRONG_STEPS * 1e3
        # like for like: ONE 800x800 frame on this rank alone, wall clock, in this same run
        fence()
        single_ms = None
        if rank == 0:
            for _ in range(2):
                h.render_batch(one, H, W)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(3):
                h.render_batch(one, H, W)
            torch.cuda.synchronize()
            single_ms = (time.perf_counter() - t2) / 3 * 1e3
        fence()
        strong = {"workload": f"C4: ONE 800x800 frame as {world} row tiles ({H // world * W} rays per rank) + RCCL gather",
                  "ms_per_frame": ms_frame, "frames_per_s": 1e3 / ms_frame, "ray_samples_per_s": H * W * (NS + NI) / ms_frame * 1e3,
                  "kernel_ms_slowest_rank": float(st[1].item()), "gather_ms_slowest_rank": float(st[2].item()), "steps": STRONG_STEPS,
                  "single_gpu_frame_ms_this_run": single_ms,
                  "speedup_vs_single_gpu_frame": None if single_ms is None else single_ms / ms_frame,
                  "note": "wall clock on both sides: one frame on rank 0 alone vs the same frame sharded over all ranks incl. the gather "
                          "(gather_ms = events around dist.gather + reassembly on the stream it runs on); ideal speedup = n_gpus"}
        # what the job actually ran on: every rank reports its device
        prop = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": local_rank, "device": prop.name, "pci_bus_id": getattr(prop, "pci_bus_id", None),
                "uuid": str(getattr(prop, "uuid", "")), "cus": prop.multi_processor_count}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=host_group)
        rccl = {"world": dist.get_world_size(), "backend": dist.get_backend(), "devices": everyone,
                "distinct_devices": len({(d["pci_bus_id"], d["uuid"], d["local_rank"]) for d in everyone})}

    if rank == 0:
        r0 = h.renderer
        flops_per_launch = (H * W * frames_per_step // world) * (NS * r0.flops_per_eval(0) + (NS + NI) * r0.flops_per_eval(1))
        achieved = flops_per_launch / (k_ms * 1e-3) / 1e12
        passes = {"f16x3": 3, "f16x1": 1, "f32": 1}[args.precision]
        # executed MFMA FLOPs: one v_mfma_f32_32x32x16_f16 (32 768 FLOP, 32 rays) per (hi, lo) tile pair of the packed stream
        # and pass, i.e. padding (63 -> 64, 27 -> 32, head rows) and the folded feature layer included as executed
        mfma_per_eval = [r0.packed_stream(w).size // 2048 for w in (0, 1)]
        executed = (H * W * frames_per_step // world) / 32 * (NS * mfma_per_eval[0] + (NS + NI) * mfma_per_eval[1]) * passes * 32768
        exec_tflops = executed / (k_ms * 1e-3) / 1e12 if args.precision != "f32" else None
        # the launches of a frame, timed apart by the library (an event between them): under the hybrid plan the DOMINANT kernel
        # is the packets instantiation over the full rounds of workgroups, the sample-split instantiation renders the rest
        launches = None
        if parts:
            flops_per_ray = NS * r0.flops_per_eval(0) + (NS + NI) * r0.flops_per_eval(1)
            launches = []
            for i, what in enumerate(("packets (SPLIT = false): the dominant kernel", "sample split (SPLIT = true): the ragged last round")[:len(parts[0])]):
                ms_i = float(np.mean([p[i][0] for p in parts]))
                rays_i = parts[0][i][1]
                launches.append({"instantiation": what if len(parts[0]) == 2 else "one launch", "rays": rays_i, "kernel_ms": ms_i,
                                 "algorithmic_tflops": rays_i * flops_per_ray / (ms_i * 1e-3) / 1e12,
                                 "frac": rays_i * flops_per_ray / (ms_i * 1e-3) / 1e12 / PEAK_F16_TFLOPS})
        traffic, traffic_note = profiled_traffic() if (world == 1 and args.precision == "f16x3" and not args.unfolded) else (None, "not profiled for this mode")
        line = {
            "metric": "ray-samples/sec (800x800, 192 samples, 8x256 MLP)", "value": value, "unit": "ray-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16x3": "f16x3 (split-fp16 MFMA, fp32 accumulate)", "f16x1": "f16", "f32": "f32"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "C3: 800x800 view, 64 coarse + 128 importance samples, 8x256 coarse + fine NeRF MLP, "
                                   "hfov 90, near 0.1 far 10, random-init weights (seeds 1000/1001)",
                       "frames_per_step": frames_per_step, "rays_per_step": rays_per_step,
                       "parallelism": f"row-tile x{world}" + ((" + RCCL gather" if backend == "nccl" else f" + {backend} gather (REHEARSAL: ranks share devices)") if world > 1 else "")},
            "per_gpu_ray_samples_per_s": value / world,
            "rays_per_s": rays_per_step * args.steps / elapsed,
            "mlp_evals_per_s": rays_per_step * evals_per_ray * args.steps / elapsed,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F16_TFLOPS,
                         "traffic": traffic, "traffic_note": traffic_note + "; algorithmic 1.76e7 B per launch (poses in, weights once, rgb/depth/acc out)",
                         "kernel": ("render_mfma_kernel<256,8,4,%s>" % ("unfolded" if args.unfolded else "folded")) if args.precision != "f32" else "render_f32_kernel",
                         "launch_plan": r0.debug_last_plan(),
                         "launches": launches,
                         "launch_plan_note": "0 = one launch of 4-packet workgroups; 2 = the full rounds of workgroups as packets + the ragged last "
                                             "round sample-split in a second launch of the same kernel template right behind it; kernel_ms covers both",
                         "kernel_ms": k_ms, "algorithmic_flops_per_launch": flops_per_launch,
                         "executed_mfma_passes": passes, "mfma_per_eval_and_pass": mfma_per_eval,
                         "executed_mfma_tflops": exec_tflops,
                         "frac_executed": None if exec_tflops is None else exec_tflops / PEAK_F16_TFLOPS,
                         # what a bare dependent chain of this MFMA sustains on RANDOM operands under the socket power cap
                         # (tools/ubench/mfma_power.hip, profiles/r01_ubench_mfma_power.txt: 1.57-1.75 GHz at ~1300 W)
                         "power_capped_mfma_peak": POWER_CAPPED_F16_TFLOPS,
                         "frac_executed_of_power_capped_peak": None if exec_tflops is None else exec_tflops / POWER_CAPPED_F16_TFLOPS,
                         # torch.matmul (hipBLASLt) fp16 8192^3 on random operands, same pool (profiles/r01_gemm_reference.txt)
                         "library_gemm_f16_sustained": LIBRARY_GEMM_F16_TFLOPS},
        }
        if strong is not None:
            line["strong"] = strong
            line["rccl"] = rccl
            line["in_process"] = in_process_probe(args, world)
        if world == 1 and not args.no_configs:
            line["configs"] = other_configs(args)
        if world == 1 and not args.no_cpu_baseline:
            rays, ref, dt, start, threads, dt_c1 = cpu_baseline(sd_c, sd_f, poses[0])
            cpu_value = CPU_SAMPLE_RAYS * (NS + NI) / dt
            line["cpu_baseline"] = {"value": cpu_value, "unit": "ray-samples/s", "cores": threads, "kind": "port",
                                    "cpu_model": cpu_model(), "logical_cpus_visible": os.cpu_count(),
                                    "sample": f"{CPU_SAMPLE_RAYS} rays (rows {start // W}..) of the same frame = {CPU_SAMPLE_RAYS // CPU_CHUNK} reference "
                                              f"chunks of {CPU_CHUNK} rays, oracle/nerf_oracle.py on torch CPU fp32, {threads} threads (= the cores this "
                                              f"process may use: affinity mask capped by the cgroup CPU quota), {dt:.1f} s; C1 (64x64, 32 samples, 4x128) in full: {dt_c1:.2f} s",
                                    "mlp_evals_per_s": CPU_SAMPLE_RAYS * evals_per_ray / dt, "rays_per_s": CPU_SAMPLE_RAYS / dt,
                                    "frame_s_extrapolated": H * W * (NS + NI) / cpu_value,
                                    "c1_frame_s": dt_c1, "c1_mlp_evals_per_s": 64 * 64 * 32 / dt_c1}
            # image quality of the timed frame against the oracle on the same rays (metric: "PSNR vs ref")
            got = frame["rgb"][0].reshape(-1, 3)[start:start + CPU_SAMPLE_RAYS].cpu().numpy()
            mse = float(np.mean((got.astype(np.float64) - ref["rgb_fine"].numpy().astype(np.float64)) ** 2))
            err = np.abs(got - ref["rgb_fine"].numpy()).max(-1)
            line["psnr_vs_oracle_db"] = 99.0 if mse == 0 else -10.0 * np.log10(mse)
            # attribution, ray by ray (tests/test_gpu_parity.py::test_c3_subset_against_golden does this on the golden subset):
            # re-render the sample's rows with the sample depths as an output and compare them with the oracle's
            rows = (start // W, (start + CPU_SAMPLE_RAYS + W - 1) // W)
            chk = h.render_batch(poses[:1], H, W, rows=rows, outputs=("rgb", "z_fine"))
            assert torch.equal(chk["rgb"].reshape(-1, 3)[:CPU_SAMPLE_RAYS], frame["rgb"][0].reshape(-1, 3)[start:start + CPU_SAMPLE_RAYS])
            dz = (chk["z_fine"].reshape(-1, NS + NI)[:CPU_SAMPLE_RAYS].cpu() - ref["z_fine"]).abs().max(-1).values.numpy()
            cliff = ref["raw_fine"][:, -1, 3].abs().numpy() < 1e-5
            same = (dz <= 2e-5) & ~cliff
            line["rgb_abs_err_vs_oracle"] = {
                "median": float(np.median(err)), "p99": float(np.quantile(err, 0.99)), "share_above_1e-4": float((err > 1e-4).mean()),
                "rays": CPU_SAMPLE_RAYS, "rays_with_the_oracles_sample_depths": int(same.sum()),
                "max_err_on_those": float(err[same].max()), "above_1e-4_on_those": int((err[same] > 1e-4).sum()),
                "above_1e-4_with_moved_depths": int(((err > 1e-4) & ~same).sum()), "cliff_rays": int(cliff.sum()),
                "note": "random unrelated coarse/fine nets: the reference's inverse-CDF sampling amplifies the ~5e-7 difference "
                        "between two fp32 evaluations of the coarse MLP by up to 1e4 in depth (DESIGN.md section 6); every ray "
                        "above 1e-4 is one whose sample depths moved, every ray sampled where the oracle sampled it is within 1e-4"}
            # the parity gate of the timed frame: a regression FAILS the run (no line) instead of printing a number beside it.
            # Bounds as in tests/test_gpu_parity.py::test_c3_subset_against_golden (measured 0.5-0.8 % / 6.5e-4 / 94-97 dB).
            q = line["rgb_abs_err_vs_oracle"]
            problems = []
            if q["above_1e-4_on_those"] != 0:
                problems.append(f"{q['above_1e-4_on_those']} rays sampled where the oracle sampled them are above 1e-4 (max {q['max_err_on_those']:.2e})")
            if q["share_above_1e-4"] >= 0.012:
                problems.append(f"{q['share_above_1e-4']:.2%} of the rays above 1e-4 (bound 1.2 %)")
            if float(err[~cliff].max()) >= 1.5e-3:
                problems.append(f"max abs error {float(err[~cliff].max()):.2e} (bound 1.5e-3)")
            if line["psnr_vs_oracle_db"] <= 50.0:
                problems.append(f"PSNR {line['psnr_vs_oracle_db']:.1f} dB (bound 50)")
            if problems:
                print(json.dumps(line), file=sys.stderr, flush=True)
                raise SystemExit("bench.py: the timed frame does not match the oracle: " + "; ".join(problems))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier(group=host_group)      # ranks != 0 wait here, on the host, while rank 0's child renders on all devices
        dist.destroy_process_group()


def in_process_probe(args, n):
    """The in-process leg in a CHILD process of rank 0 (the other ranks idle on a host-side barrier meanwhile): a path that
    has only ever run with all tiles on one GPU must not be able to take the headline line down with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--in-process-probe", str(n), "--precision", args.precision]
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": f"probe exited {out.returncode} without a result", "stderr_tail": out.stderr[-400:]}
    except Exception as exc:   # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def in_process_frame(args, sd_c, sd_f, pose, n, n_visible):
    """One C3 frame through nwe_render_tiled from this process alone: n contexts on devices 0..n-1, tiles copied into the
    frame with hipMemcpyPeerAsync.  The other ranks are idle (behind the barrier) while this runs.  A frame that did not
    go through nwe_render_tiled, or a tile that did not render, is an error, not a number."""
    try:
        import nwe_amd
        devices = list(range(n))
        if n_visible < n:
            if os.environ.get("NWE_BENCH_BACKEND", "nccl") == "nccl":
                return {"skipped": f"{n_visible} devices visible to this process, {n} needed"}
            devices = [i % n_visible for i in devices]      # rehearsal: several tiles per device, same code path
        hh = nwe_amd.NeRFReplicaInferenceHandler("office_tokyo", "synthetic", precision=args.precision, devices=devices)
        hh.set_sampling(NS, NI)
        hh.initialize_models(state_dicts=(sd_c, sd_f))
        for _ in range(2):
            hh.render_batch(pose[None], H, W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STRONG_STEPS):
            out = hh.render_batch(pose[None], H, W)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / STRONG_STEPS * 1e3
        r = hh.renderer
        tile_ms = r.tile_kernel_ms()
        res = {"workload": f"ONE 800x800 frame as {n} row tiles on devices {devices} from one process (nwe_render_tiled)",
               "ms_per_frame": ms, "tile_kernel_ms": tile_ms, "tiles": len(tile_ms), "peer_access": r.peer_access(),
               "warning": r.last_warning(), "flags": int(out["flags"].item()),
               "current_device_after": torch.cuda.current_device()}
        if not r.last_tiled:
            res["error"] = "the frame did not go through nwe_render_tiled (single-context path taken)"
        elif len(tile_ms) != n or any(not (t > 0) for t in tile_ms):
            res["error"] = f"not every tile rendered: kernel ms per tile {tile_ms}"
        else:
            # the frame must be the single-context frame, bit for bit
            one = nwe_amd.NeRFReplicaInferenceHandler("office_tokyo", "synthetic", precision=args.precision, device=devices[0])
            one.set_sampling(NS, NI)
            one.initialize_models(state_dicts=(sd_c, sd_f))
            ref = one.render_batch(pose[None], H, W)
            res["equal_to_single_context_frame"] = bool(torch.equal(ref["rgb"], out["rgb"]) and torch.equal(ref["depth"], out["depth"]))
            if not res["equal_to_single_context_frame"]:
                res["error"] = "tiled frame differs from the single-context frame"
        return res
    except Exception as exc:   # noqa: BLE001 - a diagnostic leg must not take the headline number down with it
        return {"error": f"{type(exc).__name__}: {exc}"}


def other_configs(args):
    """The BASELINE.json configurations besides the headline that one GPU can run: kernel time from
    the library's HIP events, best of three launches after one warm-up; C5 = 32 poses x 800x800 in ONE launch (wall clock too).
    The poses are the bench's GUI sweep; weights from the same generator (4x128 for C1)."""
    import nwe_amd
    res = {}
    cases = [("C1", "64x64, 32 coarse samples, coarse-only 4x128", 64, 64, 32, 0, 4, 128, 1),
             ("C2", "400x400, 64 coarse samples, 8x256", 400, 400, 64, 0, 8, 256, 1),
             ("GUI", "320x240 (the YAML's image size), 64+128, 8x256", 240, 320, 64, 128, 8, 256, 1),
             ("C5_one_gpu", "32 poses x 800x800, 64+128, 8x256 in ONE launch on one GPU", 800, 800, 64, 128, 8, 256, 32)]
    for key, what, hh, ww, ns, ni, D, Wn, n_poses in cases:
        progress(f"config {key}: {what}")
        try:
            hd = nwe_amd.NeRFReplicaInferenceHandler("office_tokyo", "synthetic", precision=args.precision)
            hd.set_sampling(ns, ni)
            hd.initialize_models(state_dicts=(nwe_amd.synthetic.make_state_dict(1000, D, Wn),
                                              nwe_amd.synthetic.make_state_dict(1001, D, Wn) if ni else None))
            ps = np.stack([sweep_pose(k, n_poses) for k in range(n_poses)])
            reps = 1 if n_poses > 1 else 3
            if n_poses == 1:
                hd.render_batch(ps, hh, ww)
            ms, wall = [], []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = hd.render_batch(ps, hh, ww)
                torch.cuda.synchronize()
                wall.append((time.perf_counter() - t0) * 1e3)
                ms.append(hd.renderer.last_kernel_ms())
            k = min(ms)
            rays = n_poses * hh * ww
            evals = rays * (ns + (ns + ni if ni else 0))
            flops = rays * (ns * hd.renderer.flops_per_eval(0) + ((ns + ni) * hd.renderer.flops_per_eval(1) if ni else 0))
            res[key] = {"workload": what, "kernel_ms": k, "wall_ms": min(wall), "rays": rays,
                        "ray_samples_per_s": rays * (ns + ni) / k * 1e3, "mlp_evals_per_s": evals / k * 1e3,
                        "algorithmic_tflops": flops / k / 1e9, "frac_of_dense_f16_peak": flops / k / 1e9 / PEAK_F16_TFLOPS,
                        "plan": hd.renderer.debug_last_plan(), "flags": int(out["flags"].item())}
            if n_poses > 1:
                res[key]["frames_per_s"] = n_poses / min(wall) * 1e3
            del hd, out
        except Exception as exc:   # noqa: BLE001 - reported, the headline stands
            res[key] = {"workload": what, "error": f"{type(exc).__name__}: {exc}"}
    # the C3 frame with networks that take no view directions (rendering.use_view_dirs: False, nerf_model.py:42-43,78-79):
    # the MFMA kernel's own instantiation - trunk + one tile of _output_linear - through the ctypes wrapper directly (the
    # handler would read the flag from its YAML)
    key, what = "C3_no_view_dirs", "800x800, 64+128, 8x256 trunk + _output_linear (use_view_dirs: False)"
    progress(f"config {key}: {what}")
    try:
        from nwe_amd.handler import pinhole_intrinsics
        r = nwe_amd.Renderer(0)
        r.set_network(0, nwe_amd.synthetic.make_state_dict(1000, 8, 256, use_view_dirs=False))
        r.set_network(1, nwe_amd.synthetic.make_state_dict(1001, 8, 256, use_view_dirs=False))
        r.set_sampling(64, 128)
        fx, fy, cx, cy = pinhole_intrinsics(800, 800)
        ms = []
        for _ in range(3):
            out = r.render(sweep_pose(0, 1), 800, 800, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, precision=args.precision)
            ms.append(r.last_kernel_ms())
        k = min(ms[1:])
        flops = 640000 * (64 * r.flops_per_eval(0) + 192 * r.flops_per_eval(1))
        res[key] = {"workload": what, "kernel_ms": k, "rays": 640000, "ray_samples_per_s": 640000 * 192 / k * 1e3,
                    "mlp_evals_per_s": 640000 * 256 / k * 1e3, "algorithmic_tflops": flops / k / 1e9,
                    "frac_of_dense_f16_peak": flops / k / 1e9 / PEAK_F16_TFLOPS, "plan": r.debug_last_plan(), "flags": int(out["flags"].item())}
        r.close()
    except Exception as exc:   # noqa: BLE001
        res[key] = {"workload": what, "error": f"{type(exc).__name__}: {exc}"}
    return res


if __name__ == "__main__":
    main()
