#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on the einstein-scan configuration.

Workload (the configuration the headline metric is quoted on): the scan stand-in mesh
(icosphere level 8, 1,310,720 faces; the real einstein.obj is not available), Emitter +
4096^2 bilinear image texture, orthographic camera of example/scan, 3840x2160, 1024 rays per
pixel, fast mode.  One "step" renders `--spp-step` consecutive sample indices of every pixel of
the frame (default 1024: a step is the whole frame, one launch of the primary kernel per rank);
the rows of the frame are cut into 16-row bands dealt round-robin to the ranks, every rank keeps
the whole scene in its own HBM, and the per-rank films are gathered onto rank 0 with one RCCL
gather after the last step.

Prints ONE JSON line on rank 0.  `value` is TotalRays (the reference's RenderStats counter,
render.go:143) of all ranks divided by the slowest rank's wall time for the K timed steps, with
EVERY tile of the frame traced on the GPU: the library's shortcut for tiles that cannot see the
scene (it counts their rays on the host without generating them) is switched off for the timed
steps, so that every ray in `value` is a ray a kernel generated and traced, as every ray of the
CPU baseline is.  The same steps with the shortcut on (the library's default) are run afterwards
and reported as `config.value_with_culling` (reference-counted rays per second, i.e. the frame
rate a caller sees) and `config.value_traced_rays` (only the rays a kernel traced, per second).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, GB/s


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--spp-step", type=int, default=1024, help="sample indices rendered per step (default: all of the frame's, one launch)")
    p.add_argument("--level", type=int, default=8, help="icosphere level of the stand-in mesh (8 = 1.31 M faces)")
    p.add_argument("--dx", type=int, default=3840)
    p.add_argument("--dy", type=int, default=2160)
    p.add_argument("--spp", type=int, default=1024, help="RaysPerPixel of the configuration")
    p.add_argument("--tex", type=int, default=4096)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-culling-leg", action="store_true", help="skip the extra, untimed steps with the screen-bounds culling on")
    p.add_argument("--no-cold-frame", action="store_true", help="skip the cold-frame measurement on a fresh scene handle")
    p.add_argument("--culling", action="store_true", help="run the TIMED steps with the culling on (profiling aid; `value` then counts host-settled rays)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work of the baseline sample")
    p.add_argument("--emulate-world", type=int, default=0,
                   help="single process renders only the row bands that rank 0 of this many ranks would own (scaling rehearsal)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo + --same-gpu rehearses the multi-rank path with every rank on GPU 0 of a one-GPU machine")
    p.add_argument("--same-gpu", action="store_true")
    p.add_argument("--check", action="store_true", help="compare a crop of the GPU image with the oracle")
    return p.parse_args()


def cpu_baseline(scene, opts, target_seconds):
    """Times the CPU oracle (the restatement of the reference algorithm) on a bounded sample of
    the same workload: every 8th tile row and column of the frame (1/64 of the tiles), with as
    many sample indices as fit the time budget.  Also returns the reference-order visit counts
    (boxes and faces tested per mesh ray) that price the algorithmic bytes of a ray."""
    from oracle import binding as ob
    from raytrace_amd import phys
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    with ob.OracleScene(scene) as orc:
        probe = phys.RenderOptions(opts.Seed, opts.RaysPerPixel, opts.MaxRayDepth, opts.Dx, opts.Dy)
        ro = phys.render_opts(probe, sample_begin=0, sample_count=1)
        img, st, ex, _ = _oracle_render(orc, scene, probe, ro, cores, 8)
        rate = st.total_rays / max(st.render_seconds, 1e-9)
        n_samples = int(max(1, min(opts.RaysPerPixel, target_seconds * rate / max(st.total_rays, 1))))
        ro = phys.render_opts(probe, sample_begin=0, sample_count=n_samples)
        img, st, ex, _ = _oracle_render(orc, scene, probe, ro, cores, 8)
    return {
        "value": st.total_rays / st.render_seconds / 1e6,
        "unit": "Mrays/s",
        "cores": int(ex.threads_used),
        "kind": "port",
        "sample": f"C++ float64 restatement of the reference algorithm (-O2, no FMA), 16x16 tile queue, every 8th tile row and "
                  f"column of the {opts.Dx}x{opts.Dy} frame x {n_samples} sample indices = {st.total_rays} rays in "
                  f"{st.render_seconds:.2f} s",
        "boxes_per_ray": ex.boxes_tested / max(ex.mesh_rays, 1),
        "faces_per_ray": ex.prims_tested / max(ex.mesh_rays, 1),
        "miss_fraction": st.rays_left_scene / max(st.total_rays, 1),
    }


def _oracle_render(orc, scene, options, ro, cores, tile_step, film=None):
    import numpy as np
    from oracle import binding as ob
    from raytrace_amd import abi, phys
    img = np.zeros((options.Dy, options.Dx, 4), dtype=np.uint8)
    st = abi.Stats()
    ex = ob.RenderExtra(cores, tile_step, film.ctypes.data if film is not None else None)
    cd = phys.camera_desc(scene.Camera[0])
    rc = orc.lib.oracle_render(orc.handle, C.byref(cd), C.byref(ro), img.ctypes.data_as(C.POINTER(C.c_uint8)), options.Dx * 4,
                               C.byref(st), C.byref(ex))
    assert rc == 0
    return img, st, ex, None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    import torch.distributed as dist

    if a.same_gpu:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    from raytrace_amd import _lib, abi, phys, scenes, shard

    lib = _lib.load(local_rank)
    scene = scenes.scan(level=a.level, dx=a.dx, dy=a.dy, spp=a.spp, depth=10, seed=0, tex_size=a.tex)
    o = scene.RenderOptions
    t0 = time.time()
    dev = phys.DeviceScene(scene)
    upload_s = time.time() - t0
    info = dev.bvh_info(0)
    cd = phys.camera_desc(scene.Camera[0])
    film = torch.zeros((o.Dy, o.Dx, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step(k, culling=False):
        s0 = (k * a.spp_step) % o.RaysPerPixel
        cnt = min(a.spp_step, o.RaysPerPixel - s0)
        ro = phys.render_opts(o, abi.MODE_FAST, rank, a.emulate_world or world, stream, s0, cnt)
        st = abi.Stats()
        if s0 == 0:
            film.zero_()  # a new frame begins (part of the step, as in rt_render)
        # RT_NO_PIXEL_CULL is read by the library at every call: set = every tile is traced
        if culling:
            os.environ.pop("RT_NO_PIXEL_CULL", None)
        else:
            os.environ["RT_NO_PIXEL_CULL"] = "1"
        _lib.check(lib.rt_render_film(dev.handle, C.byref(cd), C.byref(ro), C.c_void_p(film.data_ptr()), C.byref(st)), lib)
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def leg(culling, steps, warmup, combine):
        """`warmup` untimed steps, then `steps` timed ones between two fences.  Returns the whole-job totals
        (sums over ranks) and the slowest rank's wall time."""
        for k in range(warmup):
            step(k, culling)
        if combine and warmup > 0:
            # the first gather of a process group sets up its point-to-point connections (the barrier's all-reduce
            # does not): that one-off cost belongs to the warm-up, like the first launch of a kernel
            shard.combine_films(film)
        film.zero_()
        fence()
        t_begin = time.perf_counter()
        acc = dict(rays=0.0, trav_s=0.0, launches=0.0, trav_rays=0.0, culled=0.0)
        for k in range(steps):
            st = step(k, culling)
            acc["rays"] += st.total_rays
            acc["trav_s"] += st.traverse_seconds
            acc["launches"] += st.traverse_launches
            acc["trav_rays"] += st.traverse_rays
            acc["culled"] += st.culled_rays
        if combine:
            shard.combine_films(film)  # the per-GPU films hold disjoint row bands: one sum onto rank 0 assembles the frame
        fence()
        elapsed = time.perf_counter() - t_begin
        mine = dict(acc)
        tot = torch.tensor([acc["rays"], elapsed, acc["trav_s"], acc["launches"], acc["trav_rays"], acc["culled"]], dtype=torch.float64, device="cuda")
        tot, mx = shard.combine_counters(tot)
        return dict(rays=float(tot[0]), elapsed=float(mx[1]), trav_rays=float(tot[4]), culled=float(tot[5]), mine=mine)

    main_leg = leg(a.culling, a.steps, a.warmup, True)
    elapsed = main_leg["elapsed"]  # the slowest rank's wall time
    rays_all = main_leg["rays"]
    trav_s, trav_launches, trav_rays = main_leg["mine"]["trav_s"], main_leg["mine"]["launches"], main_leg["mine"]["trav_rays"]
    out_img = None
    spp_done = min(a.steps * a.spp_step, o.RaysPerPixel)
    if rank == 0:
        out_img = np.zeros((o.Dy, o.Dx, 4), dtype=np.uint8)
        _lib.check(lib.rt_film_resolve(C.c_void_p(film.data_ptr()), o.Dx, o.Dy, max(spp_done, 1),
                                       out_img.ctypes.data_as(C.POINTER(C.c_uint8)), o.Dx * 4, None), lib)
    film_main = film.clone() if (a.check and rank == 0) else None
    cull_leg = None
    if not a.no_culling_leg and not a.culling:
        # The same steps with the library's default schedule (tiles that cannot see the scene are settled by the host),
        # outside the timed region.  One warm-up step: it holds the pilot pass and the occupancy map.
        cull_leg = leg(True, a.steps, 1, False)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = rays_all / elapsed / 1e6
    result = {
        "metric": "Mrays/sec (whole node), einstein-scan mesh 4K",
        "value": value,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / max(a.steps, 1),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"example/scan stand-in: displaced icosphere {info.n_faces} faces (einstein.obj is not in the reference tree), "
                        f"Emitter + {a.tex}x{a.tex} bilinear image texture, orthographic camera, {o.Dx}x{o.Dy} @ {o.RaysPerPixel} spp, "
                        f"MaxRayDepth {o.MaxRayDepth}; one step = {a.spp_step} sample indices of every pixel",
            "mode": "fast (float32 conservative box culling, float64 face intersection)",
            "parallelism": f"16-row bands round-robin over {world} GPU(s), scene replicated, one RCCL gather of each GPU's rows",
            "timed_steps": "every tile traced on the GPU (RT_NO_PIXEL_CULL=1): value counts only rays a kernel generated and traced"
                           if not a.culling else "library default: tiles that cannot see the scene are settled by the host",
            "bvh_nodes": int(info.n_nodes), "bvh_depth": int(info.max_depth), "bvh_build_s": info.build_seconds,
            "scene_upload_s": upload_s,
            "rays_per_step_all_gpus": rays_all / max(a.steps, 1),
            "rank0_kernel_ms_per_step": 1e3 * float(trav_s) / max(a.steps, 1),
            "culled_ray_fraction_timed_steps": main_leg["culled"] / max(rays_all, 1.0),
        },
    }
    if cull_leg is not None:
        # camera rays of 8x8 tiles that lie outside the screen outline of the scene: counted like the reference counts
        # them (TotalRays, RaysLeftScene), settled without being generated
        result["config"]["value_with_culling"] = cull_leg["rays"] / cull_leg["elapsed"] / 1e6
        result["config"]["value_traced_rays"] = cull_leg["trav_rays"] / cull_leg["elapsed"] / 1e6
        result["config"]["ms_per_step_with_culling"] = 1e3 * cull_leg["elapsed"] / max(a.steps, 1)
        result["config"]["culled_ray_fraction"] = cull_leg["culled"] / max(cull_leg["rays"], 1.0)
    base = None
    if not a.no_cpu_baseline:
        base = cpu_baseline(scene, o, a.cpu_seconds)
        result["cpu_baseline"] = {k: base[k] for k in ("value", "unit", "cores", "kind", "sample")}
    # Roofline of the dominant kernel, k_primary<fast, mesh-only> (camera ray + closest-hit traversal + emitter
    # shading, one launch per step), over rank 0's launches of the timed steps.  Algorithmic bytes of a ray =
    #   32 (ray in) + 32 * boxes tested + 48 * faces tested + 16 (hit out)     traversal, reference visit order
    #   + 16 (film accumulate per sample) + 16 * P(hit) (4 RGBA8 texels per shaded hit)
    # with the box / face counts of the REFERENCE-order traversal measured by the CPU oracle on its sample of
    # the same frame; the launch time is the library's HIP-event time around the launches on their stream.
    # `traffic`, `hbm_counter_frac` and the `roofline_valu` block come from hardware counters, which cannot be read
    # inside this process: they are STORED values of the rocprofv3 passes over this same command
    # (tools/profile_pmc.sh -> tools/make_counters.py -> profiles/counters.json), not measured in this run.
    stored = None
    cpath = os.path.join(ROOT, "profiles", "counters.json")
    if os.path.exists(cpath):
        with open(cpath) as f:
            stored = json.load(f)
    if base is not None and trav_launches > 0:
        hit_fraction = 1.0 - base["miss_fraction"]
        bytes_per_ray = 32.0 + 32.0 * base["boxes_per_ray"] + 48.0 * base["faces_per_ray"] + 16.0 + 16.0 + 16.0 * hit_fraction
        my_rays = float(trav_rays)
        my_secs = float(trav_s)
        achieved = bytes_per_ray * my_rays / max(my_secs, 1e-12) / 1e9
        avg_launch_s = my_secs / max(float(trav_launches), 1.0)
        traffic = stored.get("k_primary_hbm_bytes_per_launch") if stored else None
        result["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": "stored profile value (profiles/counters.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command), not measured in this run",
            "hbm_counter_frac": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "kernel": "k_primary<fast, mesh-only>", "bytes_per_ray": bytes_per_ray, "boxes_per_ray": base["boxes_per_ray"],
            "faces_per_ray": base["faces_per_ray"], "rays_per_launch": my_rays / max(float(trav_launches), 1.0),
            "avg_launch_ms": 1e3 * avg_launch_s, "miss_fraction": base["miss_fraction"],
            "note": "algorithmic bytes price every box and face test of the reference's left-first walk as an HBM access; the kernel "
                    "walks the mesh once per PIXEL (the bundle of the pixel's parallel rays; its samples then test only the faces listed) and the scene (about 340 MB) sits in "
                    "L2 and the 256 MB memory-side cache, so frac exceeds 1 and says nothing about HBM: hbm_counter_frac is the "
                    "measured HBM share, and the bound that limits the kernel is vector instruction issue (roofline_valu)",
        }
        if stored and "k_primary_valu_busy" in stored:
            result["roofline_valu"] = {
                "bound": "valu", "achieved": stored["k_primary_valu_busy"], "peak": 1.0, "unit": "fraction of SIMD issue cycles",
                "frac": stored["k_primary_valu_busy"],
                "lanes_enabled": stored.get("k_primary_lanes_enabled"),
                "valu_insts_per_traced_ray": stored.get("k_primary_valu_insts_per_ray"),
                "source": "stored profile value (profiles/counters.json): 4 x SQ_ACTIVE_INST_VALU / (SIMDs x kernel cycles) of the timed launches of this command under rocprofv3 --pmc",
            }
    if a.check:  # radiance sums of the timed steps against the CPU oracle, on every 16th tile row and column
        from oracle import binding as ob
        with ob.OracleScene(scene) as orc:
            probe = phys.RenderOptions(o.Seed, o.RaysPerPixel, o.MaxRayDepth, o.Dx, o.Dy)
            ro = phys.render_opts(probe, sample_begin=0, sample_count=spp_done)
            ref_film = np.zeros((o.Dy, o.Dx, 3), dtype=np.float64)
            ref, _, _, _ = _oracle_render(orc, scene, probe, ro, 16, 16, ref_film)
        mask = ref[..., 3] == 255
        got = film_main.cpu().numpy()[mask]
        want = ref_film[mask]
        err = np.abs(got - want) / np.maximum(np.abs(want), 1.0)
        result["check"] = {"pixels": int(mask.sum()), "samples_per_pixel": int(spp_done), "max_rel_diff_of_sums": float(err.max()),
                           "frac_rel_diff_gt_1e-9": float((err > 1e-9).mean()), "nonzero_fraction": float((want > 0).any(axis=1).mean())}
    if world == 1 and not a.emulate_world and not a.no_cold_frame:
        # One whole frame on a FRESH scene handle through rt_render, as a caller that renders a single frame gets it:
        # pilot pass, occupancy map and schedule, every pass of the frame, quantisation and the copy to the host.
        os.environ.pop("RT_NO_PIXEL_CULL", None)
        t0 = time.perf_counter()
        dev2 = phys.DeviceScene(scene)
        t1 = time.perf_counter()
        art = dev2.render(scene.Camera[0], o, abi.MODE_FAST)
        t2 = time.perf_counter()
        dev2.close()
        result["config"]["cold_frame_s"] = t2 - t1
        result["config"]["cold_frame_with_scene_create_s"] = t2 - t0
        result["config"]["cold_frame_rays"] = int(art.Stats.TotalRays)
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()
    dev.close()


if __name__ == "__main__":
    main()
