#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the wavefront path tracer on BASELINE.json configs[1]
(Cornell box + Lambertian BSDF, 1920x1080, PTMIS, radius-0 sensor), one rank per GPU.

A step is one pass of the hot path over one batch: ITER_PER_STEP iterations (samples per pixel) of the full frame on
each rank.  With the defaults (--steps 16, 16 iterations per step) one GPU renders the configuration's 256 spp.
Ranks shard the *iterations* of the frame (rank r renders iterations [r*K*I, (r+1)*K*I), every (pixel, iteration)
is independent); the scene is replicated, there is no data-path collective, and the per-rank films
{sum w*rgb, sum w} are summed once per run with an RCCL reduce over xGMI.  Scaling is weak: per-GPU work is fixed.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITER_PER_STEP = 16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters-per-step", type=int, default=ITER_PER_STEP)
    ap.add_argument("--scene", default="cornell", choices=["cornell", "cornell_mesh", "cornell_glass", "outdoor"],
                    help="cornell = BASELINE.json configs[1] (default), cornell_mesh = configs[2]; cornell_glass and outdoor "
                         "(configs[4]'s ingredients: glass, metal presets, 1 M-triangle mesh, image-lit canopy + sun) run the "
                         "POOL kernels")
    ap.add_argument("--material", default="lambert", choices=["lambert", "substitute"])
    ap.add_argument("--fuse", type=int, default=1, choices=[0, 1],
                    help="zygpu option fuse_trace_shade: wf_trace_shade (closest hit + first shading pass in one kernel, default) "
                         "or wf_intersect and wf_shade apart")
    ap.add_argument("--nee", default="in-place", choices=["in-place", "staged"],
                    help="zygpu option nee_in_place: wf_scatter (default) or the wf_nee / wf_shadow rounds + wf_extend")
    ap.add_argument("--precise", type=int, default=1, choices=[0, 1],
                    help="zygpu option precise: 1 = the reference's arithmetic bit for bit (verification mode, film identical to "
                         "the oracle's), 0 = relaxed arithmetic (native divide / sqrt, contraction, OCML f32 functions) within the "
                         "RMS <= 1e-3 contract")
    ap.add_argument("--batch-mpaths", type=int, default=0, help="zygpu option batch_mpaths (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    n = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n > 1 and world != n:
        raise SystemExit(f"--gpus {n} needs WORLD_SIZE={n} (launch with torch.distributed.run)")
    dist = None
    if n > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import zyg_amd
    from zyg_amd import build as zbuild, scenes
    if not zbuild.up_to_date():
        if rank == 0:
            zbuild.build(verbose=False)
        if dist:
            dist.barrier()

    w, h, ips, k, wu = args.width, args.height, args.iters_per_step, args.steps, args.warmup
    total_spp = n * k * ips
    eng = zyg_amd.Engine()
    eng.set_device(local_rank)
    mesh = scenes.heightfield_mesh() if args.scene == "cornell_mesh" else None
    # cornell_mesh = BASELINE.json configs[2]: the 999 698-triangle heightfield with one grey Substitute material
    mesh_material = scenes.substitute((0.5, 0.5, 0.5)) if mesh is not None else None
    if args.scene == "outdoor":
        scenes.outdoor(eng, w, h, total_spp, integrator="PTMIS", material="substitute", sky="canopy", sky_map=scenes.sky_image(1024, 1024),
                       sky_spectrum=(1.0, 1.0, 1.0), showcase=True, mesh=scenes.heightfield_mesh(), max_depth=16)
    else:
        glass = {"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3), "sphere": scenes.glass(1.5)} if args.scene == "cornell_glass" else None
        scenes.cornell(eng, w, h, total_spp, integrator="PTMIS", material=args.material, boxes=mesh is None, mesh=mesh,
                       mesh_material=mesh_material, glass_objects=glass)
    eng.start_frame(0)
    lib, ctx = eng.lib, eng.context()
    if args.batch_mpaths:
        assert lib.zygpu_set_option(ctx, b"batch_mpaths", args.batch_mpaths) == 0
    assert lib.zygpu_set_option(ctx, b"fuse_trace_shade", args.fuse) == 0
    assert lib.zygpu_set_option(ctx, b"precise", args.precise) == 0, lib.zygpu_last_error().decode()
    fused = bool(args.fuse) and args.scene in ("cornell", "cornell_mesh")  # POOL scenes always run the kernels apart
    nee_in_place = args.nee == "in-place"
    assert lib.zygpu_set_option(ctx, b"nee_in_place", 1 if nee_in_place else 0) == 0

    film = torch.zeros((h, w, 4), dtype=torch.float32, device=device)
    assert lib.zygpu_set_external_film(ctx, ctypes.c_void_p(film.data_ptr())) == 0

    def render(first_iteration, count):
        rc = lib.zygpu_render(ctx, 0, first_iteration, count, 0, h)
        if rc != 0:
            raise RuntimeError("zygpu_render: " + lib.zygpu_last_error().decode())

    # algorithmic bytes per sample from the device traversal counters of one stats-enabled iteration
    lib.zygpu_enable_stats(ctx, 1)
    lib.zygpu_reset_stats(ctx)
    render(0, 1)
    assert lib.zygpu_synchronize(ctx) == 0
    st = eng.stats()
    lib.zygpu_enable_stats(ctx, 0)
    samples_stat = max(st.camera_samples, 1)
    bytes_sample_total = st.algorithmic_bytes(0) / samples_stat
    # split of the §8(d) terms over the wavefront kernels (DESIGN.md "Algorithmic bytes"); B_trav is counted per kernel
    # (zygpu_get_kernel_stats): wf_intersect walks for extension rays, wf_shade for the unoccluding lights, wf_shadow
    # for shadow rays.
    #   per extension ray, 224 = ray read 32 + hit write 16 (wf_intersect) + hit read 16 + state read 64 (wf_shade)
    #                            + state write 64 + ray write 32 (wf_extend)
    #   per shadow ray,     96 = ray write 32 + contribution write 16 (wf_nee) + ray read 32 + visibility 16 (wf_shadow)
    #   B_film 32 per camera sample: staged by finalize, gathered by film_accumulate (not a wavefront kernel)
    names = ("wf_intersect", "wf_shade", "wf_nee", "wf_shadow", "wf_extend", "wf_scatter")
    ks = (zyg_amd.Stats * len(names))()
    assert lib.zygpu_get_kernel_stats(ctx, ks) == 0

    def trav(x):
        return 32 * x.nodes_fetched + 48 * x.triangles_tested + 104 * x.instances_tested

    kernel_bytes_sample = {
        "wf_intersect": (48 * st.extension_rays + trav(ks[0])) / samples_stat,
        "wf_shade": (80 * st.extension_rays + trav(ks[1])) / samples_stat,
        "wf_nee": (48 * st.shadow_rays) / samples_stat,
        "wf_shadow": (48 * st.shadow_rays + trav(ks[3])) / samples_stat,
        "wf_extend": (96 * st.extension_rays) / samples_stat,
        "wf_scatter": (96 * st.extension_rays + 96 * st.shadow_rays + trav(ks[5])) / samples_stat,
    }
    if nee_in_place:
        kernel_bytes_sample["wf_nee"] = kernel_bytes_sample["wf_shadow"] = kernel_bytes_sample["wf_extend"] = 0.0
    if fused:  # wf_trace_shade is timed as kernel 0 and does both kernels' work; kernel 1 is wf_shade's second pass only
        kernel_bytes_sample["wf_intersect"] += kernel_bytes_sample["wf_shade"]
        kernel_bytes_sample["wf_shade"] = 0.0

    for i in range(wu):
        render(i * ips, ips)
    assert lib.zygpu_synchronize(ctx) == 0
    film.zero_()
    dtmp = ctypes.c_double * len(names)
    ltmp = ctypes.c_uint64 * len(names)
    lib.zygpu_kernel_time(ctx, dtmp(), ltmp(), 1)

    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    base = rank * k * ips
    for s in range(k):
        render(base + s * ips, ips)
    assert lib.zygpu_synchronize(ctx) == 0
    if dist:
        dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms, launches = dtmp(), ltmp()
    lib.zygpu_kernel_time(ctx, ms, launches, 0)

    out = None
    if rank == 0:
        samples_total = float(w) * h * ips * k * n
        samples_rank = float(w) * h * ips * k
        value = samples_total / elapsed / 1e6
        kernels = [(nm, ms[i], launches[i], kernel_bytes_sample[nm]) for i, nm in enumerate(names)]
        dom = max(kernels, key=lambda t: t[1])
        avg_ms = dom[1] / max(dom[2], 1)
        bytes_per_launch = dom[3] * samples_rank / max(dom[2], 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM-side bytes per launch of that kernel from the committed rocprofv3 --pmc passes of this same configuration
        # (tools/pmc_traffic.py -> profiles/traffic.json; FETCH_SIZE and WRITE_SIZE need separate passes, so they cannot
        # be collected from inside this process); null for any other configuration
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        default_cfg = (args.scene == "cornell" and args.material == "lambert" and (w, h, ips) == (1920, 1080, ITER_PER_STEP)
                       and not args.batch_mpaths)
        if default_cfg and os.path.exists(tpath):
            prof = json.load(open(tpath))
            for key, val in prof.items():
                if key.startswith(dom[0]) and "true" not in key:
                    traffic = round(val["hbm_bytes_per_launch"], 1)
        # a measured ceiling beside the nominal peak: device-to-device copy of 2 GiB, read + written bytes
        copy_gbs = None
        try:
            src = torch.empty(1 << 29, dtype=torch.float32, device=device)
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize(device)
            copy_gbs = round(4 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del src, dst
        except RuntimeError:
            pass
        weights = film[..., 3]
        geometry = "2 boxes" if mesh is None else "999698-triangle heightfield mesh (Substitute)"
        if args.scene == "cornell_glass":
            geometry = "tall box, glass box (ior 1.5, absorbing), glass Sphere"
        bsdf = "Lambertian" if args.material == "lambert" else "Substitute (roughness 0.8)"
        out = {
            "metric": "Msamples/sec (whole node) at 1080p/256spp; RMS vs CPU reference",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": n,
            "steps": k,
            "warmup": wu,
            "ms_per_step": round(elapsed / k * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"outdoor scene (ground, 2 boxes (one glass), glass + gold Spheres, 999698-triangle mesh in "
                                    f"polished iron, 1024^2 canopy image light + Distant sun), {w}x{h}, PTMIS depth 16, Sobol, box "
                                    f"sensor, {ips} iterations per step, {k * ips} spp per GPU" if args.scene == "outdoor" else
                                    f"Cornell box (5 Rectangle walls, {geometry}, 2 unoccluding Rectangle lights) + "
                                    f"{bsdf} BSDF, {w}x{h}, PTMIS depth 16, Sobol, box sensor, {ips} iterations per step, "
                                    f"{k * ips} spp per GPU"), "scene": args.scene, "material": args.material,
                       "parallelism": f"iterations x{n}"},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "measured_copy_gbs": copy_gbs,
                         "algorithmic_bytes_per_launch": round(bytes_per_launch, 1),
                         "avg_launch_ms": round(avg_ms, 5), "launches": int(dom[2]),
                         "algorithmic_bytes_per_sample": round(bytes_sample_total, 1),
                         "kernel_bytes_per_sample": round(dom[3], 1),
                         "kernel_ms": {nm: round(ms[i], 3) for i, nm in enumerate(names)},
                         "kernel_bytes": {nm: round(v, 1) for nm, v in kernel_bytes_sample.items()}},
            "film_weight_min_max": [float(weights.min().item()), float(weights.max().item())],
        }
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["rms_vs_oracle"] = cpu_baseline(eng, w, h, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    lib.zygpu_set_external_film(ctx, None)
    eng.release()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(eng, w, h, budget_s):
    """Times the CPU oracle (a port of the reference's loop; the Zig reference cannot be built here) on the host cores
    over a bounded sample of the same workload, and checks the GPU image against it on exactly those iterations."""
    import subprocess

    import numpy as np

    import zyg_amd
    oracle_so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(oracle_so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    oracle = ctypes.CDLL(oracle_so)
    oracle.zyg_oracle_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                         ctypes.c_void_p]
    oracle.zyg_oracle_resolve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    sd = eng.lib.su_ext_scene_desc().contents
    vd = eng.lib.su_ext_view_desc().contents
    cores = os.cpu_count() or 1
    film = np.zeros((h, w, 4), dtype=np.float32)
    st = zyg_amd.Stats()
    iters = 0
    t0 = time.perf_counter()
    while True:
        oracle.zyg_oracle_render(ctypes.addressof(sd), ctypes.addressof(vd), 0, iters, 1, 1, 0, h, film.ctypes.data, cores,
                                 ctypes.byref(st))
        iters += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or dt / iters * (iters + 1) > budget_s * 1.5:
            break
    cpu_val = w * h * iters / dt / 1e6

    # parity on the same iterations: GPU film for iterations [0, iters) vs the oracle film
    lib, ctx = eng.lib, eng.context()
    lib.zygpu_set_external_film(ctx, None)
    lib.zygpu_clear_film(ctx, ctypes.c_float(0.0))
    lib.zygpu_render(ctx, 0, 0, iters, 0, h)
    lib.zygpu_synchronize(ctx)
    gpu_img = eng.resolve(w, h)
    cpu_img = np.zeros_like(film)
    oracle.zyg_oracle_resolve(ctypes.addressof(vd), film.ctypes.data, cpu_img.ctypes.data, w * h)
    a = np.clip(np.nan_to_num(gpu_img[..., :3]), 0, 1).astype(np.float64)
    b = np.clip(np.nan_to_num(cpu_img[..., :3]), 0, 1).astype(np.float64)
    rms = float(np.sqrt(np.mean((a - b) ** 2)))
    gpu_film = eng.download_film(w, h)
    identical = float(np.mean(np.all(gpu_film == film, axis=-1)))
    base = {"value": round(cpu_val, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{iters} iteration(s) of the same {w}x{h} frame ({w * h * iters} camera samples), {dt:.1f} s"}
    return base, {"rms": rms, "iterations": iters, "bit_identical_pixels": identical}


if __name__ == "__main__":
    main()
