#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the wavefront path tracer on BASELINE.json configs[1]
(Cornell box + Lambertian BSDF, 1920x1080, PTMIS, radius-0 sensor), one rank per GPU.

A step is one pass of the hot path over one batch: ITER_PER_STEP iterations (samples per pixel) of the full frame on
each rank.  With the defaults (--steps 16, 16 iterations per step) one GPU renders the configuration's 256 spp.
Ranks shard the *iterations* of the frame (zyg_amd.distributed.ShardedFrame: rank r renders iterations
[r*K*I, (r+1)*K*I), every (pixel, iteration) is independent); the scene is replicated, there is no data-path
collective, and the per-rank films {sum w*rgb, sum w} are summed once per run with an RCCL reduce over xGMI.  Scaling is
weak by default: per-GPU work is fixed.  `--spp-total N` fixes the whole job instead (strong scaling: BASELINE.json configs[3],
1024 spp split over the ranks), sharded by `--shard iterations` (rank r renders N / world iterations of every row) or
`--shard rows` (rank r renders all N iterations of its rows: the tile-sharded form, exact for the radius-0 sensor).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITER_PER_STEP = 16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
KERNELS = ("wf_intersect", "wf_shade", "wf_nee", "wf_shadow", "wf_extend", "wf_scatter")

# The bytes of SURVEY §8(d)'s formula that have to cross HBM whatever the caches do: the path state / ray / hit records of
# every extension ray (B_q = 224), the shadow ray records (B_sq = 96) and the film tap (B_film = 32 at radius 0).  B_trav (the
# BVH nodes, triangles and instances a ray touches) is the cache-resident part for scenes that fit the 256 MB Infinity Cache.
B_Q, B_SQ, B_FILM = 224, 96, 32


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters-per-step", type=int, default=ITER_PER_STEP)
    ap.add_argument("--scene", default="cornell", choices=["cornell", "cornell_mesh", "cornell_mound", "cornell_glass", "outdoor", "outdoor_mound"],
                    help="cornell = BASELINE.json configs[1] (default), cornell_mesh = configs[2] (SURVEY 8(d): the closed displaced "
                         "sphere of 999 698 triangles hanging at 60 %% of the box's height), cornell_mound = the lighter variant "
                         "(the heightfield on the floor); cornell_glass and outdoor (configs[4]'s ingredients: glass, metal presets, "
                         "1 M-triangle mesh, image-lit canopy + sun) run the POOL kernels")
    ap.add_argument("--material", default="lambert", choices=["lambert", "substitute", "procedural"],
                    help="procedural: Substitute whose colours, roughness and normals are procedural textures (Checker, Noise, Mix: wf_textures)")
    ap.add_argument("--box-triangles", type=int, default=12, choices=[12, 18, 48, 108, 972],
                    help="triangles of each of the two Cornell boxes (12 = BASELINE.json configs[1]); the others measure what a scene "
                         "just past the kernels' thresholds costs (meshes of at most 16 triangles are tested from the constant block)")
    ap.add_argument("--extra-props", type=int, default=0, help="small Rectangle plaques on the back wall (the scene has 7 solid props without them; the prop tables hold 112)")
    ap.add_argument("--fuse", type=int, default=1, choices=[0, 1],
                    help="zygpu option fuse_trace_shade: wf_trace_shade (closest hit + first shading pass in one kernel, default) "
                         "or wf_intersect and wf_shade apart")
    ap.add_argument("--nee", default="in-place", choices=["in-place", "staged"],
                    help="zygpu option nee_in_place: wf_scatter (default) or the wf_nee / wf_shadow rounds + wf_extend")
    ap.add_argument("--precise", type=int, default=0, choices=[0, 1],
                    help="zygpu option precise: 1 = the reference's arithmetic bit for bit (verification mode: the film is "
                         "identical to the oracle's), 0 = relaxed arithmetic (native divide / sqrt, contraction, OCML f32 "
                         "functions), inside the contract's RMS <= 1e-3 (default)")
    ap.add_argument("--fuse-tick", type=int, default=0, choices=[0, 1],
                    help="zygpu option fuse_tick: wf_scatter of vertex k and wf_trace_shade of vertex k + 1 in one launch (wf_tick)")
    ap.add_argument("--persistent", type=int, default=0, choices=[0, 1],
                    help="zygpu option persistent: one persistent kernel per batch, path state in LDS (wf_persistent)")
    ap.add_argument("--mesh-kernels", type=int, default=1, choices=[0, 1],
                    help="zygpu option mesh_kernels: the large meshes' walks in persistent kernels whose lanes fetch their own rays")
    ap.add_argument("--mesh-closest", type=int, default=0, choices=[0, 1], help="zygpu option mesh_closest (wf_mesh_closest)")
    ap.add_argument("--bvh8", type=int, default=1, choices=[0, 1], help="zygpu option bvh8: the persistent mesh walkers read the eight-wide compressed trees (wf_shadow_u8, wf_mesh_closest8)")
    ap.add_argument("--mesh-closest8", type=int, default=1, choices=[0, 1], help="zygpu option mesh_closest8: with bvh8, the closest hits of the rays that enter a large mesh in the persistent walker too")
    ap.add_argument("--sort-alive", type=int, default=-1, choices=[-1, 0, 1],
                    help="zygpu option sort_alive: the surviving paths in one sub-queue per hit prop, so that the consumer's waves stand on one "
                         "surface (-1: the library's choice, on in the scenes of the mesh kernels)")
    ap.add_argument("--pack-prev", type=int, default=1, choices=[0, 1],
                    help="zygpu option pack_prev: the previous vertex's origin / normal in dead words of the path's first line (0: two records of the second line)")
    ap.add_argument("--hit-by-queue", type=int, default=1, choices=[0, 1],
                    help="zygpu option hit_by_queue: the hit records at their vertex' place in the alive queue (dense) instead of by path slot")
    ap.add_argument("--compact-blocks", type=int, default=0, choices=[0, 1],
                    help="zygpu option compact_blocks: with pack_prev, path blocks 128 B apart (only their first line is used) instead of 256")
    ap.add_argument("--regenerate", type=int, default=0, choices=[0, 1],
                    help="zygpu option regenerate: the regenerating kernels (a small pool whose slots are all alive, plain record arrays, no queues)")
    ap.add_argument("--pair-small", type=int, default=1, choices=[0, 1],
                    help="zygpu option pair_small: parallelogram halves of a small mesh as one item of the small-triangle table")
    ap.add_argument("--regen-kslots", type=int, default=0, help="zygpu option regen_kslots: slots of that pool in units of 1024 (0 = library default, 2048)")
    ap.add_argument("--batch-mpaths", type=int, default=0, help="zygpu option batch_mpaths (0 = library default)")
    ap.add_argument("--shard", default="iterations", choices=["iterations", "rows"],
                    help="how the ranks split the frame: its iterations (every rank renders all rows) or its rows (every rank renders all "
                         "iterations of its own rows; always strong scaling: the frame's work is fixed)")
    ap.add_argument("--spp-total", type=int, default=0,
                    help="strong scaling: the whole job's samples per pixel, split over the ranks (and over --steps); 0 = weak scaling, "
                         "--steps x --iters-per-step per rank")
    ap.add_argument("--spawn", type=int, default=0, choices=[0, 1],
                    help="`--gpus N` started without a launcher: 0 (default) = ONE process drives the N devices through a ZygpuGroup "
                         "(su_ext_set_devices: a host thread and a context per device, ncclReduce of the films from C++); 1 = this process "
                         "starts N ranks, one process per GPU, as torch.distributed.run would")
    ap.add_argument("--reduce", default="group", choices=["group", "torch"],
                    help="who sums the ranks' films: group = zygpu_group_reduce (ncclReduce called from C++ behind the C ABI, default); "
                         "torch = torch.distributed.reduce on the film tensor (the cross-check of rounds 1-4)")
    ap.add_argument("--print-config", action="store_true", help="print bench_config() (the key a committed profile is quoted under) and exit")
    ap.add_argument("--pool-relaxed", type=int, default=1, choices=[0, 1],
                    help="zygpu option pool_relaxed: with --precise 0 the scenes with Glass run the relaxed units like every other scene (default); "
                         "0: they get the precise units whatever --precise says (the rule until round 4)")
    ap.add_argument("--split-materials", type=int, default=1, choices=[0, 1],
                    help="zygpu option split_materials: wf_scatter_h per material class over the sorted alive queue's sub-queues (default) or the Substitute instantiation for all")
    ap.add_argument("--lazy-head", type=int, default=1, choices=[0, 1],
                    help="zygpu option lazy_head: the first shading pass leaves a vertex' sampler and throughput to wf_scatter (default) or writes them back")
    ap.add_argument("--shadow-u-blocks", type=int, default=4, help="zygpu option shadow_u_blocks: blocks of the persistent wf_shadow_u per CU")
    ap.add_argument("--fuse-pool", type=int, default=1, choices=[0, 1],
                    help="zygpu option fuse_pool: the POOL scenes' closest hit + first shading pass in one kernel (default) or as wf_intersect + wf_shade")
    ap.add_argument("--split-nee", type=int, default=1, choices=[0, 1],
                    help="zygpu option split_nee: the POOL scenes' complete wf_scatter as wf_light (no closure) + wf_scatter<.., PHASE 2> (default) or in one kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the timed oracle sample (and again of the parity band)")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[2] (1 M-triangle mesh) measurement in `extra`")
    args = ap.parse_args()
    # strong scaling: the job's total spp is fixed, every rank renders its share of it in --steps steps
    args.strong = bool(args.spp_total) or "rows" == args.shard
    if args.spp_total:
        per_rank = args.spp_total // (args.gpus if "iterations" == args.shard else 1)
        if per_rank * (args.gpus if "iterations" == args.shard else 1) != args.spp_total or per_rank % args.steps:
            ap.error(f"--spp-total {args.spp_total} must split evenly over {args.gpus} rank(s) and {args.steps} steps")
        args.iters_per_step = per_rank // args.steps
    return args


def build_scene(eng, scenes, name, material, w, h, total_spp, box_triangles=12, extra_props=0):
    # BASELINE.json configs[2]: 999 698 triangles with one grey Substitute material; cornell_mesh: SURVEY 8(d)'s closed displaced
    # sphere centred at 60 % of the box's height (most paths and shadow rays cross it), cornell_mound: the heightfield on the floor
    mesh = scenes.displaced_sphere_mesh() if name == "cornell_mesh" else scenes.heightfield_mesh() if name == "cornell_mound" else None
    mesh_position = scenes.MESH_CENTRE if name == "cornell_mesh" else (0, 0.02, 0.1)
    mesh_material = scenes.substitute((0.5, 0.5, 0.5)) if mesh is not None else None
    if name in ("outdoor", "outdoor_mound"):
        # (box_lift: the boxes stand 1 mm above the ground: a glass box ON it has its bottom face in the ground's plane, and which of two
        # coincident surfaces a ray meets first hangs on the last bit; zygpu.hip relaxedUnits)
        # outdoor: configs[4]'s "same ingredients": the mesh is configs[2]'s closed displaced sphere, standing just above the ground behind the boxes
        sphere = "outdoor" == name
        scenes.outdoor(eng, w, h, total_spp, integrator="PTMIS", material="substitute", showcase=True,
                       mesh=scenes.displaced_sphere_mesh() if sphere else scenes.heightfield_mesh(), mesh_position=(0.0, 0.8, -1.6) if sphere else (0.0, 0.01, -1.6),
                       max_depth=16, sky_entity={"parameters": {"sun": {"rotation": [-55, 20, 0]}}}, box_lift=1e-3)
    else:
        glass = {"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3), "sphere": scenes.glass(1.5)} if name == "cornell_glass" else None
        overrides = None
        if "procedural" == material:  # every surface with procedural textures: a filtered checker, octaves of noise, a stochastic Mix, a noise normal map
            S, noise = scenes.substitute_full, lambda **kw: {"Noise": kw}
            checker = lambda scale, a, b: {"Checker": {"scale": scale, "colors": [list(a), list(b)]}}  # noqa: E731
            overrides = {"floor": S(checker(6.0, (0.9, 0.85, 0.8), (0.15, 0.1, 0.1)), roughness=noise(scale=[7, 5, 1], levels=3, attenuation=0.6)),
                         "back": S({"Mix": {"a": checker(3.0, (0.1, 0.6, 0.2), (0.8, 0.8, 0.2)), "b": [0.25, 0.3, 0.7], "weight": noise(scale=[3, 3, 3], transition=0.1)}},
                                   roughness=0.4, normal=noise(scale=[11, 11, 11], levels=2, attenuation=1.0)),
                         "left": S(checker(5.0, (0.8, 0.2, 0.2), (0.9, 0.7, 0.6)), roughness=0.6),
                         "right": S(noise(scale=[9, 9, 9], levels=4, attenuation=1.0), roughness=0.5),
                         "tall": S(noise(scale=[5, 5, 5], levels=4, attenuation=1.7), roughness=0.25, metallic=1.0, normal=noise(scale=[13, 13, 13])),
                         "short": S(checker(4.0, (0.9, 0.9, 0.9), (0.2, 0.3, 0.8)), roughness=noise(scale=[3, 3, 3]))}
            material = "substitute"
        scenes.cornell(eng, w, h, total_spp, integrator="PTMIS", material=material, boxes=mesh is None, mesh=mesh,
                       mesh_material=mesh_material, glass_objects=glass, mesh_position=mesh_position, overrides=overrides,
                       box_subdivide={12: 1, 18: 1, 48: 2, 108: 3, 972: 9}[box_triangles], box_first_face=(1, 4) if 18 == box_triangles else None,
                       extra_props=extra_props, box_lift=1e-3 if glass else 0.0)


def describe(name, material, w, h, ips, k, relaxed):
    """`relaxed`: the arithmetic the library really ran (zygpu_get_option "relaxed_units": a scene with Glass gets the precise
    units whatever --precise says)."""
    mode = "relaxed arithmetic (RMS <= 1e-3 contract)" if relaxed else "precise arithmetic (film bit-identical to the CPU oracle)"
    if name in ("outdoor", "outdoor_mound"):
        which = "closed displaced sphere (configs[2]'s mesh)" if "outdoor" == name else "heightfield mesh on the ground (the lighter variant of rounds 3-4)"
        return (f"outdoor scene (ground, 2 boxes (one glass) 1 mm above it, glass + gold Spheres, 999698-triangle {which} in polished iron, the Sky "
                f"entity: sun with its radiance table + 1024^2 canopy image, both the procedural stand-in for the Prague bake), {w}x{h}, PTMIS depth 16, Sobol, box sensor, {ips} iterations per step, "
                f"{k * ips} spp per GPU, {mode}")
    geometry = {"cornell": "2 boxes", "cornell_mesh": "999698-triangle closed displaced sphere (Substitute) centred at 60 % of the box's height",
                "cornell_mound": "999698-triangle heightfield mesh (Substitute) on the floor",
                "cornell_glass": "tall box, glass box (ior 1.5, absorbing), both 1 mm above the floor, glass Sphere"}[name]
    bsdf = ("Lambertian" if material == "lambert" else "Substitute with procedural textures on every surface (Checker, Noise, Mix, a noise normal map)"
            if material == "procedural" else "Substitute (roughness 0.8)")
    return (f"Cornell box (5 Rectangle walls, {geometry}, 2 unoccluding Rectangle lights) + {bsdf} BSDF, {w}x{h}, PTMIS depth 16, "
            f"Sobol, box sensor, {ips} iterations per step, {k * ips} spp per GPU, {mode}")


class Measurement:
    """One scene on this rank's GPU: the stats iteration (algorithmic bytes), warm-up, the timed steps."""

    def __init__(self, args, zyg_amd, scenes, torch, frame, device, scene, material, steps, warmup, size=None, group_spec=None):
        """group_spec: None (one context; the ranks' films, if any, are summed by torch.distributed), {"devices": [ids]} (this process is
        the whole world: one ZygpuGroup over the listed devices) or {"devices": [id], "rank": r, "world": n, "dist": dist} (one process per
        GPU: the ranks join one RCCL communicator made in C++, its id carried by torch.distributed); include/zygpu.h "Device groups"."""
        self.args, self.zyg_amd, self.torch, self.frame, self.device = args, zyg_amd, torch, frame, device
        self.group_spec = group_spec
        self.scene, self.material, self.steps, self.warmup = scene, material, steps, warmup
        self.w, self.h = size or (args.width, args.height)
        self.ips = args.iters_per_step
        self.eng = zyg_amd.Engine()
        self.eng.log_to(sys.stderr)  # stdout carries the one JSON line and nothing else
        self.eng.set_device(device.index)
        self.world = frame.world
        if group_spec:
            self.eng.set_devices(group_spec["devices"], 0 if "iterations" == frame.mode else 1, 1)  # (flag 1: RCCL also for a world of one)
            if "world" in group_spec:
                ident = [self.eng.unique_id() if 0 == group_spec["rank"] else None]
                group_spec["dist"].broadcast_object_list(ident, src=0)
                self.eng.set_world(ident[0], group_spec["rank"], group_spec["world"])
            else:
                self.world = len(group_spec["devices"])
        main = scene == args.scene  # the variants' knobs are the main scene's
        build_scene(self.eng, scenes, scene, material, self.w, self.h, (self.world if "iterations" == frame.mode else 1) * steps * self.ips,
                    args.box_triangles if main else 12, args.extra_props if main else 0)
        self.eng.start_frame(0)
        if not args.pair_small:  # read when the scene is uploaded (the context exists from the first frame on): upload again
            self.eng.set_option("pair_small", 0)
            self.eng.start_frame(0)
        self.lib, self.ctx = self.eng.lib, self.eng.context()
        self.group = self.eng.group()
        assert (self.group is not None) == bool(group_spec)
        if args.batch_mpaths:
            self.eng.set_option("batch_mpaths", args.batch_mpaths)
        defaults = parse_defaults()
        # the shipped library's options, then the ones only variant builds know (ZYG_AMD_LIB=variants/lib_all.so): named when they are
        # not at their default, so that a default run never asks for one
        for option, arg in (("fuse_trace_shade", "fuse"), ("pool_relaxed", "pool_relaxed"), ("shadow_u_blocks", "shadow_u_blocks"), ("precise", "precise"),
                            ("mesh_kernels", "mesh_kernels"), ("bvh8", "bvh8"), ("sort_alive", "sort_alive")):
            self.eng.set_option(option, getattr(args, arg))
        for option in ("split_nee", "fuse_pool", "lazy_head", "split_materials", "persistent", "fuse_tick", "mesh_closest", "mesh_closest8", "regenerate",
                       "pack_prev", "compact_blocks", "hit_by_queue"):
            if getattr(args, option) != defaults[option]:
                self.eng.set_option(option, getattr(args, option))
        if args.regen_kslots:
            self.eng.set_option("regen_kslots", args.regen_kslots)
        self.eng.set_option("nee_in_place", 1 if args.nee == "in-place" else 0)
        self.pool_mode = bool(self.eng.get_option("pool_mode"))
        self.relaxed = bool(self.eng.get_option("relaxed_units"))  # what really runs: Glass scenes get the precise units
        self.fused = bool(args.fuse) and not self.pool_mode  # POOL scenes always run the kernels apart
        self.film = torch.zeros((self.h, self.w, 4), dtype=torch.float32, device=device)
        assert self.lib.zygpu_set_external_film(self.ctx, ctypes.c_void_p(self.film.data_ptr())) == 0

    def render(self, first_iteration, count, y_begin, y_end):
        if 0 != self.lib.zygpu_render(self.ctx, 0, first_iteration, count, y_begin, y_end):
            raise RuntimeError("zygpu_render: " + self.lib.zygpu_last_error().decode())

    def synchronize(self):
        self.eng.synchronize()  # (every device of a group)
        self.torch.cuda.synchronize(self.device)

    def render_world(self, step):
        """Step `step` of the whole job through the group: the world's iterations [step * W * ips, (step + 1) * W * ips) (or, sharded by
        rows, the step's ips iterations of every row), of which every rank renders its share (zygpu_group_render)."""
        shard = 0 if "iterations" == self.frame.mode else 1
        count = self.world * self.ips if 0 == shard else self.ips
        if 0 != self.lib.zygpu_group_render(self.group, 0, step * count, count, 0, self.h, shard):
            raise RuntimeError("zygpu_group_render: " + self.lib.zygpu_last_error().decode())

    def my_rows(self):
        """rows [y0, y1) this rank's (first) context renders in a step"""
        if self.group and "rows" == self.frame.mode:
            first, length = ctypes.c_uint32(0), ctypes.c_uint32(0)
            rank0 = self.group_spec.get("rank", 0)
            assert 0 == self.lib.zygpu_group_share(self.h, self.world, rank0, ctypes.byref(first), ctypes.byref(length))
            return first.value, first.value + length.value
        _, _, y0, y1 = self.frame.step_range(0, self.ips, self.steps, self.h)
        return y0, y1

    def run(self):
        lib, ctx, zyg_amd = self.lib, self.ctx, self.zyg_amd
        # algorithmic bytes per sample from the device traversal counters of one stats-enabled iteration
        # (they equal the oracle's counters exactly: tests/test_gpu_parity.py)
        lib.zygpu_enable_stats(ctx, 1)
        lib.zygpu_reset_stats(ctx)
        self.render(0, 1, 0, self.h)
        self.synchronize()
        st = self.eng.stats()
        lib.zygpu_enable_stats(ctx, 0)
        ks = (zyg_amd.Stats * len(KERNELS))()
        assert lib.zygpu_get_kernel_stats(ctx, ks) == 0
        n_stat = float(max(st.camera_samples, 1))

        def trav(x):
            return 32 * x.nodes_fetched + 48 * x.triangles_tested + 104 * x.instances_tested

        # split of the §8(d) terms over the wavefront kernels (DESIGN.md "Algorithmic bytes"); B_trav is counted per kernel:
        #   per extension ray, 224 = ray read 32 + hit write 16 (wf_intersect) + hit read 16 + state read 64 (wf_shade)
        #                            + state write 64 + ray write 32 (wf_extend / wf_scatter)
        #   per shadow ray,     96 = ray write 32 + contribution write 16 (wf_nee) + ray read 32 + visibility 16 (wf_shadow)
        #   B_film 32 per camera sample: staged by finalize, gathered by film_accumulate (not a wavefront kernel)
        kb = {"wf_intersect": (48 * st.extension_rays + trav(ks[0])) / n_stat,
              "wf_shade": (80 * st.extension_rays + trav(ks[1])) / n_stat,
              "wf_nee": (48 * st.shadow_rays) / n_stat,
              "wf_shadow": (48 * st.shadow_rays + trav(ks[3])) / n_stat,
              "wf_extend": (96 * st.extension_rays) / n_stat,
              "wf_scatter": (96 * st.extension_rays + 96 * st.shadow_rays + trav(ks[5])) / n_stat}
        # the same split for the bytes that must cross HBM (no B_trav)
        sb = {"wf_intersect": 48 * st.extension_rays / n_stat, "wf_shade": 80 * st.extension_rays / n_stat,
              "wf_nee": 48 * st.shadow_rays / n_stat, "wf_shadow": 48 * st.shadow_rays / n_stat,
              "wf_extend": 96 * st.extension_rays / n_stat, "wf_scatter": (96 * st.extension_rays + 96 * st.shadow_rays) / n_stat}
        if self.args.nee == "in-place":
            for d in (kb, sb):
                d["wf_nee"] = d["wf_shadow"] = d["wf_extend"] = 0.0
        if self.fused:  # wf_trace_shade is timed as kernel 0 and does both kernels' work; kernel 1 is wf_shade's second pass only
            for d in (kb, sb):
                d["wf_intersect"] += d["wf_shade"]
                d["wf_shade"] = 0.0

        for i in range(self.warmup):
            if self.group:
                self.render_world(i)  # (every device allocates its pool and warms up)
            else:
                self.render(i * self.ips, self.ips, 0, self.h)
        self.synchronize()
        if self.group:
            assert 0 == lib.zygpu_group_clear_film(self.group, 0.0)  # every member's film (rank 0's is self.film)
            self.synchronize()
        self.film.zero_()
        dtmp, ltmp = ctypes.c_double * len(KERNELS), ctypes.c_uint64 * len(KERNELS)
        lib.zygpu_kernel_time(ctx, dtmp(), ltmp(), 1)

        def body():
            if self.group:
                # the path's one exchange step behind the C ABI: ncclReduce of the ranks' films onto rank 0's, called from C++
                # (csrc/hip/group.hip) in stream order behind each rank's kernels; the synchronize waits for it
                for s in range(self.steps):
                    self.render_world(s)
                if 0 != lib.zygpu_group_reduce(self.group):
                    raise RuntimeError("zygpu_group_reduce: " + lib.zygpu_last_error().decode())
                self.eng.synchronize()
                return
            for s in range(self.steps):
                self.frame.render_step(self.render, s, self.ips, self.steps, self.h)
            self.eng.synchronize()
            if self.frame.world > 1 and "gloo" == self.frame.dist.get_backend():  # test set-up only: through host copies
                host = self.film.cpu()
                self.frame.reduce(host)
                self.film.copy_(host)
            else:
                self.frame.reduce(self.film)

        elapsed = self.frame.timed(body, self.synchronize)
        ms, launches = dtmp(), ltmp()
        lib.zygpu_kernel_time(ctx, ms, launches, 0)
        y0, y1 = self.my_rows()
        samples_rank = float(self.w) * (y1 - y0) * self.ips * self.steps
        samples_job = samples_rank * self.world if "iterations" == self.frame.mode else float(self.w) * self.h * self.ips * self.steps
        times = {nm: ms[i] for i, nm in enumerate(KERNELS)}
        counts = {nm: int(launches[i]) for i, nm in enumerate(KERNELS)}
        # option mesh_kernels on a scene with a large mesh: the in-place stage is wf_scatter_h (+ its resume launches, timed as
        # wf_scatter) and wf_shadow_u, which walks the meshes for the suspended shadow rays (timed as wf_shadow): one stage, one
        # share of the algorithmic bytes; a tick's launches are counted as one
        self.mesh_stage = "in-place" == self.args.nee and times["wf_shadow"] > 0.0
        if self.mesh_stage:
            times["wf_scatter"] += times["wf_shadow"]
            counts["wf_scatter"] = counts["wf_intersect"] if self.fused else counts["wf_scatter"]
            times["wf_shadow"], counts["wf_shadow"] = 0.0, 0
        kernels = [(nm, times[nm], counts[nm], kb[nm], sb[nm]) for nm in KERNELS]
        dom = max(kernels, key=lambda t: t[1])
        avg_ms = dom[1] / max(dom[2], 1)
        per_launch = samples_rank / max(dom[2], 1)
        achieved = dom[3] * per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        per_kernel = {}
        for nm, t_ms, ln, bytes_ps, state_ps in kernels:
            if ln and t_ms > 0:
                shown = "wf_trace_shade" if (self.fused and "wf_intersect" == nm) else "wf_scatter_h+wf_shadow_u" if (self.mesh_stage and "wf_scatter" == nm) else nm
                per_kernel[shown] = {"ms": round(t_ms, 3), "launches": ln, "avg_launch_ms": round(t_ms / ln, 5),
                                     "frac": round(bytes_ps * samples_rank / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                     "state_frac": round(state_ps * samples_rank / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        return {
            "per_kernel": per_kernel,
            "rays_per_sample": {"extension": round(st.extension_rays / n_stat, 3), "shadow": round(st.shadow_rays / n_stat, 3),
                                "nodes": round(st.nodes_fetched / n_stat, 2), "triangles": round(st.triangles_tested / n_stat, 2),
                                "instances": round(st.instances_tested / n_stat, 2)},
            "elapsed": elapsed, "value": samples_job / elapsed / 1e6, "dominant": dom[0],
            "avg_ms": avg_ms, "launches": dom[2], "achieved": achieved,
            "algorithmic_bytes_per_launch": dom[3] * per_launch, "state_bytes_per_launch": dom[4] * per_launch,
            "kernel_bytes_per_sample": dom[3], "kernel_state_bytes_per_sample": dom[4],
            "bytes_per_sample": st.algorithmic_bytes(0) / n_stat,
            "state_bytes_per_sample": (B_FILM * st.camera_samples + B_Q * st.extension_rays + B_SQ * st.shadow_rays) / n_stat,
            "kernel_ms": {nm: round(ms[i], 3) for i, nm in enumerate(KERNELS)},
            "mesh_stage": self.mesh_stage,
            "kernel_bytes": {nm: round(v, 1) for nm, v in kb.items()},
        }

    def close(self):
        self.lib.zygpu_set_external_film(self.ctx, None)
        self.eng.release()


def bench_config(args):
    """The arguments a committed profile must have been taken with to be quoted for this run."""
    cfg = {"scene": args.scene, "material": args.material, "size": [args.width, args.height, args.iters_per_step],
           "fuse": args.fuse, "nee": args.nee, "precise": args.precise, "batch_mpaths": args.batch_mpaths, "gpus": args.gpus}
    # every option that changes which kernels run, how they move their state, or the scene: named only when it is not at its
    # default, so that a profile of the default configuration keeps its key and any other run is refused by it
    defaults = parse_defaults()
    for key in ("regenerate", "regen_kslots", "sort_alive", "pack_prev", "compact_blocks", "hit_by_queue", "mesh_kernels", "mesh_closest",
                "bvh8", "mesh_closest8", "pair_small", "box_triangles", "extra_props", "fuse_tick", "persistent", "shard", "spp_total", "pool_relaxed", "split_nee", "fuse_pool", "shadow_u_blocks", "lazy_head", "split_materials"):
        if getattr(args, key) != defaults[key]:
            cfg[key] = getattr(args, key)
    return cfg


_DEFAULTS = None


def parse_defaults():
    global _DEFAULTS
    if _DEFAULTS is None:
        saved = sys.argv
        sys.argv = [saved[0]]
        try:
            _DEFAULTS = vars(parse_args())
        finally:
            sys.argv = saved
    return _DEFAULTS


def committed_profile(args, name, kernel, source_hash, overrides=None):
    """One kernel's entry of a committed rocprofv3 --pmc summary (profiles/<name>: traffic.json from tools/pmc_traffic.py, the
    FETCH_SIZE and WRITE_SIZE passes; pmc_sq.json from tools/pmc_summary.py, the SQ pass).  Counters need profiler passes of
    their own, so this process cannot collect them; a profile is quoted only for the bench arguments it names AND for the
    kernels' sources it was taken with (`_meta.device_hash` = zyg_amd.build.device_hash(), the contents of csrc/hip and include/): a
    profile of other kernels is refused, not reported as if it were this library's."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, {"profile": "profiles/" + name, "refused": "missing"}
    prof = json.load(open(path))
    meta = prof.get("_meta", {})
    want = bench_config(args)
    want.update(overrides or {})  # (an `extra` scene: the run's arguments with that scene and its size)
    if meta.get("config") != want:
        return None, {"profile": "profiles/" + name, "refused": "taken with other bench arguments"}
    # the kernels' sources (csrc/hip + include/: build.device_hash); a profile from before that field existed names all of csrc/
    from zyg_amd import build as zbuild
    mine, theirs = (zbuild.device_hash(), meta["device_hash"]) if "device_hash" in meta else (source_hash, meta.get("source_hash"))
    if theirs != mine:
        return None, {"profile": "profiles/" + name, "refused": f"taken with other sources ({theirs} != {mine})"}
    for key, val in prof.items():
        if key.startswith(kernel) and "true" not in key.split("<")[-1].split(",")[0]:  # the non-counting instantiation
            return val, {"profile": "profiles/" + name, "source_commit": meta.get("source_commit"), "device_hash": mine,
                         "fetch_size_factor": meta.get("fetch_size_factor")}
    return None, {"profile": "profiles/" + name, "refused": "kernel not in the profile"}


_COUNTING = ("wf_scatter<true", "wf_trace_shade<true", "wf_shade<true", "wf_intersect<true", "wf_shadow<true")  # (STATS is their first parameter)


def committed_step_traffic(args, name, source_hash, overrides, ms_per_step):
    """What a scene's whole step moves across HBM according to the committed counter passes (profiles/<name>: tools/pmc_traffic.py's
    summary of separate --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py with `steps` steps, corrected as MI355X_MICROARCH.md prescribes):
    the sum over the wavefront kernels (the counting instantiations of the statistics iteration excluded) of bytes per launch x launches,
    per step, and that over this run's measured step time and the 8 TB/s peak.  Quoted under the same rule as committed_profile: the
    same bench arguments and the same kernels' sources, else refused."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return {"source": {"profile": "profiles/" + name, "refused": "missing"}}
    prof = json.load(open(path))
    meta = prof.get("_meta", {})
    want = bench_config(args)
    want.update(overrides or {})
    if meta.get("config") != want:
        return {"source": {"profile": "profiles/" + name, "refused": "taken with other bench arguments"}}
    from zyg_amd import build as zbuild
    mine, theirs = (zbuild.device_hash(), meta["device_hash"]) if "device_hash" in meta else (source_hash, meta.get("source_hash"))
    if theirs != mine:
        return {"source": {"profile": "profiles/" + name, "refused": f"taken with other sources ({theirs} != {mine})"}}
    steps = float(meta.get("steps", 2))
    kernels = {k: v for k, v in prof.items() if k.startswith("wf_") and not k.startswith(_COUNTING)}
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in kernels.values()) / steps
    return {"traffic_bytes_per_step": round(total, 1),
            "traffic_frac_step": round(total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms_per_step else None,
            "per_kernel_gb_per_launch": {k.split("(")[0][:64]: round(v["hbm_bytes_per_launch"] / 1e9, 4) for k, v in kernels.items()},
            "source": {"profile": "profiles/" + name, "source_commit": meta.get("source_commit"), "device_hash": mine, "steps": steps}}


def headline_frac(r, traffic_gbs):
    """roofline.frac / frac_model / frac_source of one measurement (r: Measurement.run()'s result; traffic_gbs: the dominant kernel's
    counted HBM traffic per second from a committed profile, or None)."""
    seconds = r["avg_ms"] * 1e-3
    model = r["achieved"] / HBM_PEAK_GBS
    saturated = any(v["frac"] > 1.0 for v in r["per_kernel"].values())
    state = r["state_bytes_per_launch"] / seconds / 1e9 / HBM_PEAK_GBS if seconds else None
    if not saturated:
        frac, source = model, "model (SURVEY 8(d) algorithmic bytes)"
    elif traffic_gbs:
        frac, source = traffic_gbs / HBM_PEAK_GBS, "hbm.traffic_frac (counted HBM bytes of the committed profile; the model is saturated)"
    else:
        frac, source = state, "hbm.state_frac (the records that must cross HBM; the model is saturated and no committed profile matches this build)"
    return {"frac": round(frac, 5) if frac is not None else None, "frac_model": round(model, 5), "frac_source": source}


def host_cpu():
    """CPU model, logical / physical core counts, and the CPUs this process may really use (affinity and cgroup quota)."""
    model, cores = None, set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
                cores.add((phys, core))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    return {"model": model, "logical": logical, "physical": len(cores) or None, "usable": usable, "cgroup_quota": quota}


class Oracle:
    """The CPU oracle (a port of the reference's loop; the Zig reference cannot be built here).  Checker and reported
    baseline only: nothing of it is on the measured path."""

    def __init__(self, zyg_amd):
        import importlib.util
        spec = importlib.util.spec_from_file_location("oracle_build", os.path.join(ROOT, "oracle", "build.py"))
        oracle_build = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(oracle_build)  # compiled for this box's CPU (-march=native)
        self.lib = ctypes.CDLL(oracle_build.ensure_built())
        self.lib.zyg_oracle_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                               ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                               ctypes.c_void_p]
        self.lib.zyg_oracle_resolve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        self.zyg_amd = zyg_amd

    def render(self, eng, film, first, count, y0, y1, threads):
        sd, vd = eng.lib.su_ext_scene_desc().contents, eng.lib.su_ext_view_desc().contents
        st = self.zyg_amd.Stats()
        t0 = time.perf_counter()
        rc = self.lib.zyg_oracle_render(ctypes.addressof(sd), ctypes.addressof(vd), 0, first, count, 1, y0, y1, film.ctypes.data,
                                        threads, ctypes.byref(st))
        assert rc == 0
        return time.perf_counter() - t0

    def resolve(self, eng, film):
        import numpy as np
        vd = eng.lib.su_ext_view_desc().contents
        out = np.zeros_like(film)
        self.lib.zyg_oracle_resolve(ctypes.addressof(vd), film.ctypes.data, out.ctypes.data, film.shape[0] * film.shape[1])
        return out


def cpu_baseline(m, oracle, budget_s):
    """Times the oracle on the host cores over a bounded sample of the same workload: one probe iteration per candidate
    thread count (all usable CPUs / one per physical core), then ONE call of N iterations of the full frame with the better
    count, N sized to the budget."""
    import numpy as np
    w, h = m.w, m.h
    cpu = host_cpu()
    candidates = sorted({cpu["usable"], max(1, min(cpu["usable"], cpu["physical"] or cpu["usable"]))}, reverse=True)
    film = np.zeros((h, w, 4), dtype=np.float32)
    oracle.render(m.eng, film, 0, 1, 0, min(h, 64), candidates[0])  # page in the library and the film
    probes = {}
    for t in candidates:
        film[:] = 0
        probes[t] = oracle.render(m.eng, film, 0, 1, 0, h, t)
    threads = min(probes, key=probes.get)
    iters = max(1, int(budget_s / probes[threads]))
    film[:] = 0
    dt = oracle.render(m.eng, film, 0, iters, 0, h, threads)
    rate = w * h * iters / dt / 1e6
    return {"value": round(rate, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "note": f"scalar C++ port of the reference's loop (the Zig reference, @Vector(4, f32) arithmetic, cannot be built here), compiled -O3 "
                    f"-march=native on this box, {threads} thread(s) of a {cpu['usable']}-CPU quota"
                    + (f" (cgroup {cpu['cgroup_quota']:g}; the host has {cpu['logical']} logical CPUs)" if cpu.get("cgroup_quota") else ""),
            "per_thread": round(rate / threads, 5), "cpu": cpu,
            "probe_seconds_per_iteration": {str(t): round(s, 3) for t, s in probes.items()},
            "sample": f"one call, {iters} iteration(s) of the same {w}x{h} frame ({w * h * iters} camera samples), {dt:.1f} s"}, rate, threads


def parity_band(m, oracle, cpu_rate, threads, budget_s, iterations):
    """The timed render's own film against the oracle: the same `iterations` samples per pixel on a band of rows sized to
    the budget (whole 32-row tiles around the middle of the frame).  RMS over RGB of the resolved image, clamped to [0, 1]
    (SURVEY §8(d)); in the precise mode also the share of film pixels that are the same bits."""
    import numpy as np
    w, h = m.w, m.h
    rows = int(cpu_rate * 1e6 * budget_s / max(1, w * iterations))
    rows = max(8, min(h, rows // 32 * 32 if rows >= 32 else rows))
    y0 = max(0, (h - rows) // 2 // 32 * 32)
    y1 = min(h, y0 + rows)
    cpu_film = np.zeros((h, w, 4), dtype=np.float32)
    dt = oracle.render(m.eng, cpu_film, 0, iterations, y0, y1, threads)
    gpu_film = m.film.cpu().numpy()
    a = np.clip(np.nan_to_num(_resolve_rows(oracle, m, gpu_film, y0, y1)), 0, 1).astype(np.float64)
    b = np.clip(np.nan_to_num(_resolve_rows(oracle, m, cpu_film, y0, y1)), 0, 1).astype(np.float64)
    return {"rms": float(np.sqrt(np.mean((a - b) ** 2))), "iterations": iterations, "rows": [y0, y1],
            "bit_identical_pixels": float(np.mean(np.all(gpu_film[y0:y1] == cpu_film[y0:y1], axis=-1))),
            "tolerance": 1e-3, "oracle_seconds": round(dt, 1)}


def _resolve_rows(oracle, m, film, y0, y1):
    return oracle.resolve(m.eng, film)[y0:y1, :, :3]


def launch_ranks(args):
    """`bench.py --gpus N` started plainly (no launcher: WORLD_SIZE unset): this process builds the library if it has to and starts
    the N ranks as fresh child processes, one per GPU, before it has imported torch or made any HIP call itself (a process that
    has initialised the GPU is never re-executed); rank 0's child prints the one JSON line on the inherited stdout."""
    import socket
    from zyg_amd import build as zbuild
    if not os.environ.get("ZYG_AMD_LIB"):  # (a development variant is measured as it was built)
        zbuild.ensure_built()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and 0 == rc:  # one rank failed: the others would wait in a collective for ever
                    rc = code
                    for q in procs:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


def main():
    # dmabuf IPC is the only mode the host driver supports; HSA reads the flag once, when the first HIP call initialises it:
    # set it before torch is imported
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse_args()
    if args.print_config:
        print(json.dumps(bench_config(args)))
        return
    n = args.gpus
    launcher = "WORLD_SIZE" in os.environ
    if n > 1 and not launcher and args.spawn:
        sys.exit(launch_ranks(args))
    # without a launcher one process is the whole world: its ZygpuGroup holds the N devices
    one_process = n > 1 and not launcher
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n > 1 and launcher and world != n:
        raise SystemExit(f"--gpus {n} under a launcher needs WORLD_SIZE={n}, not {world}")

    # The library is built before any process group exists, under a file lock: whichever rank comes first builds it when it is
    # missing or older than its sources, the others wait and find it fresh.  Nothing is ever measured with a stale library.
    from zyg_amd import build as zbuild
    if not os.environ.get("ZYG_AMD_LIB"):  # (a development variant, tools/variants_bench.sh, is measured as it was built)
        zbuild.ensure_built()
    source_hash = zbuild.source_hash()

    import numpy as np  # noqa: F401
    import torch

    import zyg_amd
    from zyg_amd import distributed, scenes
    # ZYG_BENCH_BACKEND=gloo ZYG_BENCH_ONE_DEVICE=1: the N > 1 path of this file on a box with one GPU (tests/test_distributed.py):
    # every rank on device 0, the films summed over gloo through host copies.  The measurement runs RCCL, one rank per GPU.
    backend = os.environ.get("ZYG_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", 0 if os.environ.get("ZYG_BENCH_ONE_DEVICE") else local_rank)
    # the process group first (device_id binds the RCCL communicator to this rank's GPU), then the device
    # ZYG_BENCH_GROUP=1: a process group also for one rank, so that the collectives (the film reduce on the device tensor, the
    # barrier, the max-over-ranks) run through RCCL on a box with a single GPU
    grouped = (n > 1 and not one_process) or bool(os.environ.get("ZYG_BENCH_GROUP"))
    dist = distributed.init_process_group(backend, rank, world, device) if grouped else None
    torch.cuda.set_device(device)
    frame = distributed.ShardedFrame(rank, 1 if one_process else n, dist, args.shard, args.height)
    # The film reduce behind the C ABI (include/zygpu.h "Device groups"): this rank's device is one rank of an RCCL communicator that
    # csrc/hip/group.hip makes and reduces over; torch.distributed only carries the communicator's id (and keeps the barrier and the
    # max-over-ranks of the timing protocol).  --reduce torch (and the gloo test set-up, whose ranks share one GPU, which RCCL refuses)
    # sums the films with torch.distributed instead.
    group_spec = None
    if one_process:
        group_spec = {"devices": [0] * n if os.environ.get("ZYG_BENCH_ONE_DEVICE") else list(range(n))}
    elif grouped and "group" == args.reduce and "nccl" == backend:
        group_spec = {"devices": [device.index], "rank": rank, "world": world, "dist": dist}
    collective = None if not (grouped or one_process) else (
        ("ncclReduce from C++ (ZygpuGroup), " + ("one process, a host thread per device" if one_process else "one process per GPU")) if group_spec
        else dist.get_backend() + " (torch.distributed.reduce)")

    m = Measurement(args, zyg_amd, scenes, torch, frame, device, args.scene, args.material, args.steps, args.warmup, group_spec=group_spec)
    if group_spec and 1 != m.lib.zygpu_group_uses_rccl(m.group):
        collective = "film_add kernel (ZygpuGroup whose members share a device: a test set-up)"
    r = m.run()

    if 0 == rank:
        prof_kernel = "wf_trace_shade" if (m.fused and "wf_intersect" == r["dominant"]) else r["dominant"]
        traffic, source = committed_profile(args, "traffic.json", prof_kernel, source_hash)
        sq, sq_source = committed_profile(args, "pmc_sq.json", prof_kernel, source_hash)
        hbm_bytes = traffic["hbm_bytes_per_launch"] if traffic else None
        # a measured ceiling beside the nominal peak: the library's own float4 grid-stride copy kernel over 2 x 2 GiB
        # (zygpu_measure_copy; MI355X_MICROARCH.md measures 6.29 TB/s in that shape), read + written bytes; torch's copy_ reaches
        # ~4.9 TB/s on the same boxes and flattered `traffic_frac_of_measured_copy` until round 3
        copy_gbs = None
        gbs = ctypes.c_double(0.0)
        if 0 == m.lib.zygpu_measure_copy(m.ctx, ctypes.c_uint64(1 << 31), 4, ctypes.byref(gbs)) and gbs.value > 0:
            copy_gbs = round(gbs.value, 1)
        weights = m.film[..., 3]
        seconds = r["avg_ms"] * 1e-3
        traffic_gbs = hbm_bytes / seconds / 1e9 if hbm_bytes else None
        issue = None
        if sq:
            # SQ_ACTIVE_INST_VALU counts quad-cycles per wave instruction (MI355X_MICROARCH.md), SQ_BUSY_CYCLES clock cycles summed
            # over the 32 SQ instances (8 XCDs x 4 shader engines); 1024 SIMDs: VALU-busy share of the SIMD cycles = 4 A / (B / 32 * 1024)
            valu_busy = sq["SQ_ACTIVE_INST_VALU"] / (8.0 * sq["SQ_BUSY_CYCLES"]) if sq.get("SQ_BUSY_CYCLES") else None
            issue = {"valu_busy": round(valu_busy, 4) if valu_busy else None,
                     "valu_floor_ms": round(valu_busy * r["avg_ms"], 5) if valu_busy else None,
                     "wait_share_of_wave_cycles": round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 4) if sq.get("SQ_WAVE_CYCLES") else None,
                     "valu_instructions_per_wave": round(sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"], 1) if sq.get("SQ_WAVES") else None,
                     "vgprs": sq.get("_vgpr"), "scratch_bytes_per_lane": sq.get("_scratch"), "source": sq_source}
        else:
            issue = {"valu_busy": None, "source": sq_source}
        out = {
            "metric": "Msamples/sec (whole node) at 1080p/256spp; RMS vs CPU reference",
            "value": round(r["value"], 3),
            "unit": "Msamples/s",
            "n_gpus": n,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(r["elapsed"] / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": describe(args.scene, args.material, args.width, args.height, args.iters_per_step, args.steps, m.relaxed),
                       "scene": args.scene, "material": args.material,
                       # the arithmetic the library ran (zygpu_get_option "relaxed_units"), beside what was asked for
                       "arithmetic": "relaxed" if m.relaxed else "precise", "arithmetic_requested": "precise" if args.precise else "relaxed",
                       "parallelism": f"{args.shard} x{n}" + (f", {args.spp_total} spp in all (strong)" if args.spp_total else
                                                             ", the frame's rows split (strong)" if "rows" == args.shard else " (weak)"),
                       "spp_per_rank": args.steps * args.iters_per_step, "collective": collective},
            # What bounds the dominant kernel, most believable figure first:
            #   hbm:    the bytes that really cross HBM.  `state_frac`: the ray / hit / state records it must move (B_q, B_sq of
            #           SURVEY 8(d), no traversal data) over its duration and the peak; `traffic_frac`: the HBM-side bytes rocprofv3
            #           counted for it (committed passes of this very library, else null); `waste` = traffic / state bytes.
            #   issue:  the SIMDs' VALU-busy share from the SQ counters of the same passes: the kernel's floor if every other
            #           cycle were hidden.
            #   `achieved` / `frac`: the contract's figure, SURVEY 8(d)'s cache-less algorithmic bytes (B_trav included) over the
            #           duration.  For scenes whose traversal data sits in SGPRs / constant memory / L2 most of those bytes never
            #           reach HBM, so it can exceed 1: `model_saturated` lists the kernels where it does.
            # `frac`: where SURVEY 8(d)'s cache-less model is saturated (some kernel's model bytes over its duration exceed the peak: most
            # of them never reach HBM) the headline fraction is what does cross HBM: the counted traffic of the dominant kernel where a
            # committed profile of this very library holds it, else the bytes it must move (state records): `frac_source` says which;
            # the model's figure stays as `frac_model` / `achieved`.
            "roofline": {"bound": "hbm", "kernel": r["dominant"], "achieved": round(r["achieved"], 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", **headline_frac(r, traffic_gbs),
                         "traffic": round(hbm_bytes, 1) if hbm_bytes else None, "traffic_source": source,
                         "hbm": {"state_frac": round(r["state_bytes_per_launch"] / seconds / 1e9 / HBM_PEAK_GBS, 5) if seconds else None,
                                 "traffic_frac": round(traffic_gbs / HBM_PEAK_GBS, 5) if traffic_gbs else None,
                                 "traffic_frac_of_measured_copy": round(traffic_gbs / copy_gbs, 5) if traffic_gbs and copy_gbs else None,
                                 "waste": round(hbm_bytes / r["state_bytes_per_launch"], 3) if hbm_bytes and r["state_bytes_per_launch"] else None,
                                 "state_bytes_per_launch": round(r["state_bytes_per_launch"], 1), "measured_copy_gbs": copy_gbs,
                                 "step": committed_step_traffic(args, "traffic.json", source_hash, None, r["elapsed"] / args.steps * 1e3)},
                         "issue": issue,
                         "model_saturated": [k for k, v in r["per_kernel"].items() if v["frac"] > 1.0],
                         "per_kernel": r["per_kernel"],
                         "algorithmic_bytes_per_launch": round(r["algorithmic_bytes_per_launch"], 1),
                         "avg_launch_ms": round(r["avg_ms"], 5), "launches": r["launches"],
                         "algorithmic_bytes_per_sample": round(r["bytes_per_sample"], 1),
                         "state_bytes_per_sample": round(r["state_bytes_per_sample"], 1),
                         "kernel_bytes_per_sample": round(r["kernel_bytes_per_sample"], 1),
                         "kernel_ms": r["kernel_ms"], "kernel_bytes": r["kernel_bytes"]},
            "rays_per_sample": r["rays_per_sample"],
            "film_weight_min_max": [float(weights.min().item()), float(weights.max().item())],
            "source_hash": source_hash,
        }
        oracle = None
        if 1 == n and not args.no_cpu_baseline:
            oracle = Oracle(zyg_amd)
            out["cpu_baseline"], cpu_rate, threads = cpu_baseline(m, oracle, args.cpu_seconds)
            out["rms_vs_oracle"] = parity_band(m, oracle, cpu_rate, threads, args.cpu_seconds, args.steps * args.iters_per_step)
    m.close()

    # BASELINE.json configs[2] (the 1 M-triangle mesh in the box) in the same run, so that its number is witnessed by whoever
    # runs this file: fewer steps, same sizes, its own roofline figures and parity band.  cornell_mesh is the configuration
    # (SURVEY 8(d)'s closed displaced sphere hanging in the box), cornell_mound the lighter variant (the heightfield on the floor).
    if 0 == rank and 1 == n and not args.no_extra and args.scene == "cornell" and not args.strong:
        out["extra"] = {}
        # configs[2] and its lighter variant at the run's size; configs[4]'s scene at its own 3840 x 2160 (depth 16, the Sky entity, glass
        # and conductors: the POOL kernels) and the glass Cornell box at 1080p: each with its own roofline figures, parity band and
        # the arithmetic the library really ran
        for name, size in (("cornell_mesh", None), ("cornell_mound", None), ("outdoor", (3840, 2160)), ("outdoor_mound", (3840, 2160)), ("cornell_glass", None)):
            m2 = Measurement(args, zyg_amd, scenes, torch, frame, device, name, "lambert", max(2, args.steps // (8 if size else 4)), 1, size)
            r2 = m2.run()
            extra = {"value": round(r2["value"], 3), "unit": "Msamples/s", "steps": m2.steps, "ms_per_step": round(r2["elapsed"] / m2.steps * 1e3, 3),
                     "workload": describe(name, "lambert", m2.w, m2.h, args.iters_per_step, m2.steps, m2.relaxed),
                     "arithmetic": "relaxed" if m2.relaxed else "precise", "size": [m2.w, m2.h],
                     "kernel": r2["dominant"], "achieved": round(r2["achieved"], 2), **headline_frac(r2, None),
                     "state_frac": round(r2["state_bytes_per_launch"] / (r2["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if r2["avg_ms"] else None,
                     # the scene's own committed counter passes (profiles/traffic_<scene>.json, pmc_sq_<scene>.json), same refusal rule as the main scene's
                     "hbm": committed_step_traffic(args, f"traffic_{name}.json", source_hash, {"scene": name, "size": [m2.w, m2.h, args.iters_per_step]},
                                                   r2["elapsed"] / m2.steps * 1e3),
                     "avg_launch_ms": round(r2["avg_ms"], 5), "launches": r2["launches"], "algorithmic_bytes_per_sample": round(r2["bytes_per_sample"], 1),
                     "state_bytes_per_sample": round(r2["state_bytes_per_sample"], 1), "kernel_ms": r2["kernel_ms"],
                     "per_kernel": r2["per_kernel"], "rays_per_sample": r2["rays_per_sample"],
                     "mesh_stage": "wf_scatter = wf_scatter_h + wf_shadow_u" if r2["mesh_stage"] else None}
            if oracle is not None:
                film = np.zeros((m2.h, m2.w, 4), dtype=np.float32)
                y_mid = m2.h // 2 // 32 * 32
                probe = oracle.render(m2.eng, film, 0, 1, y_mid, y_mid + 32, threads)
                rate = m2.w * 32 / probe / 1e6
                extra["rms_vs_oracle"] = parity_band(m2, oracle, rate, threads, args.cpu_seconds / 2, m2.steps * args.iters_per_step)
            out["extra"][name] = extra
            m2.close()

    if 0 == rank:
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
