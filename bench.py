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
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, GB/s
# Vector-issue peak of the chip: 256 CUs x 4 SIMDs, one wave-wide float64 / integer vector instruction per 4 cycles and
# SIMD (16 lanes a cycle: the rate behind the 78.6 TFLOP/s float64 vector peak), at the 2.4 GHz peak clock.
SIMDS = 1024
PEAK_CLOCK_HZ = 2.4e9
VALU_PEAK_GINST = SIMDS * PEAK_CLOCK_HZ / 4.0 / 1e9  # wave instructions per second, in units of 1e9

# The fewest vector instructions a wave of 64 samples of one pixel can spend on one round of the beam kernel, by stage
# (wave instructions; what is wave-uniform is taken to cost nothing: it could sit in scalar registers).  Every float64
# operation the reference prescribes for a ray is counted once, in the reference's own arithmetic (no fused multiply-add).
# "every ray": paid by each of the wave's rounds; "hit": only by rounds whose rays hit the mesh; "listed": rounds of a
# pixel whose bundle lists at least one face (the others end after the camera ray).
VALU_FLOOR = [
    # stage, who pays, wave instructions, what they are
    ("philox", "every ray", 58, "Philox4x32-10, one block for the two jitter draws: 10 rounds x (2 multiplies 32 x 32 -> 64 + 4 two-input xors; gfx950 has no "
                                "three-input xor), less the first round's two products of wave-uniform counter words; keys bumped by the scalar unit"),
    ("draws", "every ray", 8, "two float64 in [0, 1) from four 32-bit words: 2 x (2 conversions, 1 multiply-add pair)"),
    ("jitter", "every ray", 8, "render.go:202-210: 2 x (add, exactly rounded division by Dx / Dy in 3)"),
    ("camera", "every ray", 16, "camera_orthographic.go:24-46: 2 x (subtract, multiply) + 3 x (2 multiplies, 2 adds)"),
    # (render.go:147-150, the NaN / Inf guard: nothing per ray for a camera whose constants the host has found finite and far from the end
    # of the float64 range, as this frame's are; otherwise one class test per origin component)
    ("frame", "listed", 12, "origin in the bundle's float32 frame: 3 x (multiply, add, subtract, conversion)"),
    ("pretest", "listed", 15, "one float32 face pre-test: s = o - p0 (3), two dot products (6), margins and compares (6)"),
    ("face", "hit", 36, "shape_mesh.go:95-124 for ONE face, with what does not depend on the ray (edges, d x e2, 1/a) taken as given: "
                        "s (3), u (6), s x e1 (9), v (6), t (6), six compares"),
    ("keep", "hit", 8, "closest hit kept: t, u, v, face (4 selects of 2 registers)"),
    ("attributes", "hit", 18, "shape_mesh.go:125-141 for an emitter: texture coordinates, w = 1 - u - v (2) + 2 x (3 multiplies, 2 adds); point and normal are not read"),
    ("texture", "hit", 96, "texture_image.go:87-188 bilinear, repeat: wrap and scale (12), floor / fraction (6), 4 texel addresses (8), "
                           "16 channel unpack + premultiply (32), 9 lerps x 3 (27), truncations to 16 bits (3), / 65535 exactly (9) minus alpha never read"),
    ("emit", "hit", 6, "material_emitter.go:22-25 and the pixel's running sum: 3 multiplies, 3 adds"),
    ("loop", "every ray", 2, "sample index, live mask (the round counter is scalar)"),
]


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
    p.add_argument("--check", action="store_true", help="(kept for old command lines: the check is on by default)")
    p.add_argument("--no-check", action="store_true", help="skip the comparison of the timed steps' film with the CPU oracle's")
    return p.parse_args()


def self_launch(a):
    """`bench.py --gpus N` started as ONE process (no WORLD_SIZE in the environment): this process starts the N ranks itself,
    one per GPU, through torch.distributed.run -- before it has made any GPU call -- and ends with their exit code.  The
    reference uses the whole machine from one call (render.go:297-331: runtime.NumCPU() workers)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def source_hash():
    """Identifies the device code the library was built from: what stored hardware-counter figures are tied to."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytrace_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def cpu_baseline(scene, opts, target_seconds):
    """Times the CPU oracle (the restatement of the reference algorithm) on a bounded sample of
    the same workload: every 8th tile row and column of the frame (1/64 of the tiles), with as
    many sample indices as fit the time budget.  Also returns the reference-order visit counts
    (boxes and faces tested per mesh ray) that price the algorithmic bytes of a ray."""
    import numpy as np
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
        film = np.zeros((opts.Dy, opts.Dx, 3), dtype=np.float64)  # the oracle's radiance sums: what the check compares
        img, st, ex, _ = _oracle_render(orc, scene, probe, ro, cores, 8, film)
    return {
        "film": film, "mask": img[..., 3] == 255, "n_samples": n_samples,
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
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))  # nothing in this process has touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s): the two must agree")
    import numpy as np
    import torch
    import torch.distributed as dist

    if a.same_gpu:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
    elif local_rank >= torch.cuda.device_count():  # counting devices does not initialise the GPU
        sys.exit(f"bench.py: rank {rank} wants GPU {local_rank} but this machine shows {torch.cuda.device_count()} "
                 f"(--same-gpu with --backend gloo rehearses the multi-rank path on one GPU)")
    torch.cuda.set_device(local_rank)
    ranks_in_collective = 1
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            one = torch.ones(1, dtype=torch.int32, device="cuda")
        else:
            dist.init_process_group("gloo")
            one = torch.ones(1, dtype=torch.int32)
        dist.all_reduce(one)  # the number of ranks the collective library (RCCL for nccl) actually joined
        ranks_in_collective = int(one.item())
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
    do_check = not a.no_check and not a.no_cpu_baseline
    film_main = film.clone() if (do_check and rank == 0) else None
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
            "parallelism": f"16-row bands round-robin over {world} GPU(s), one process per GPU, scene replicated, one gather of each GPU's rows "
                           f"({'RCCL' if a.backend == 'nccl' else 'gloo'})",
            "ranks_in_collective": ranks_in_collective,
            "launched_by": "bench.py itself (torch.distributed.run child of --gpus N)" if os.environ.get("TORCHELASTIC_RUN_ID") and world > 1 else "caller",
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
    if do_check:
        # The film of the TIMED steps (the kernel the bench times, at the bench's size) against the CPU oracle's radiance sums
        # on the pixels the baseline rendered: every 8th tile row and column.  A pixel's sum is what render.go:186-217 adds up
        # over its samples, before the division and the 8-bit quantisation.
        n_cmp = base["n_samples"]
        same_film = n_cmp == spp_done and not a.emulate_world
        if same_film:
            got_film = film_main
        else:
            # the oracle's time budget covered fewer sample indices than a step renders (slow host), or this process holds
            # only a share of the rows: render exactly the oracle's sample range once more, every row, default path
            got_film = torch.zeros_like(film)
            ro = phys.render_opts(o, abi.MODE_FAST, 0, 1, stream, 0, n_cmp)
            _lib.check(lib.rt_render_film(dev.handle, C.byref(cd), C.byref(ro), C.c_void_p(got_film.data_ptr()), C.byref(abi.Stats())), lib)
            torch.cuda.synchronize()
        mask = base["mask"]
        got = got_film.cpu().numpy()[mask]
        want = base["film"][mask]
        err = np.abs(got - want) / np.maximum(np.abs(want), 1.0)
        ok = bool(err.max() <= 1e-9)
        result["check"] = {
            "ok": ok, "pixels": int(mask.sum()), "samples_per_pixel": int(n_cmp), "max_rel_diff_of_sums": float(err.max()),
            "frac_rel_diff_gt_1e-12": float((err > 1e-12).mean()), "nonzero_fraction": float((want > 0).any(axis=1).mean()),
            "film": "the timed steps' own film" if same_film else "an extra render of the oracle's sample range (untimed)",
            "kernel": "k_beam" if n_cmp >= 64 else "k_primary (packet walk: fewer than 64 samples per pass)",
            "bar": "radiance sums per pixel and channel equal to 1e-9 relative (same samples, same float64 face arithmetic; only the "
                   "order of the additions differs)",
        }
    # Roofline of the dominant kernel, k_beam (camera ray + the pixel's face list + emitter shading, one launch per
    # step), over rank 0's launches of the timed steps; launch time = the library's HIP-event time around the launches on
    # their own stream.  The kernel is bound by vector instruction issue, not by HBM (the scene sits in L2 and the
    # memory-side cache), so the roofline is the chip's vector issue rate and what is priced are the instructions the work
    # NEEDS (VALU_FLOOR above), not those the kernel executes: frac = needed instructions per second / issue peak.
    stored = None
    cpath = os.path.join(ROOT, "profiles", "counters.json")
    if os.path.exists(cpath):
        with open(cpath) as f:
            stored = json.load(f)
    src = source_hash()
    if stored is not None and stored.get("csrc_sha16") != src:
        result["stored_counters"] = (f"withheld: profiles/counters.json was taken at device sources {stored.get('csrc_sha16')}, this tree is {src} "
                                     f"(rerun tools/profile_pmc.sh and tools/make_counters.py)")
        stored = None
    if trav_launches > 0:
        my_rays, my_secs = float(trav_rays), float(trav_s)
        avg_launch_s = my_secs / max(float(trav_launches), 1.0)
        hit_fraction = (1.0 - base["miss_fraction"]) if base is not None else None
        listed_fraction = None
        floor = None
        if hit_fraction is not None:
            # rounds of pixels whose bundle lists a face: every hit ray's pixel does; of the others, those on the outline.
            # Taken as the hit fraction itself (a lower bound: the floor must not be overstated).
            listed_fraction = hit_fraction
            weight = {"every ray": 1.0, "listed": listed_fraction, "hit": hit_fraction}
            floor = sum(n * weight[who] for _, who, n, _ in VALU_FLOOR)  # wave instructions per round of 64 rays
        if floor is not None:
            achieved = floor / 64.0 * my_rays / max(my_secs, 1e-12) / 1e9
            result["roofline"] = {
                "bound": "valu", "achieved": achieved, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": achieved / VALU_PEAK_GINST,
                "traffic": stored.get("hbm_bytes_per_launch") if stored else None,
                "kernel": "k_beam<image texture>", "avg_launch_ms": 1e3 * avg_launch_s, "rays_per_launch": my_rays / max(float(trav_launches), 1.0),
                "needed_valu_insts_per_64_rays": floor, "hit_fraction": hit_fraction,
                "floor_by_stage": [{"stage": st, "paid_by": who, "wave_insts": n} for st, who, n, _ in VALU_FLOOR],
                "peak_is": f"{SIMDS} SIMDs x {PEAK_CLOCK_HZ / 1e9} GHz / 4 cycles per wave-wide float64 or integer vector instruction",
                "note": "achieved = vector instructions the work needs (floor_by_stage, weighted by the fraction of rays that pay each stage) x "
                        "rays per second measured in this run; the instructions the kernel actually executes are in valu_counters",
            }
        if stored and "valu_busy" in stored:
            result["valu_counters"] = {
                "valu_busy": stored["valu_busy"], "lanes_enabled": stored.get("lanes_enabled"),
                "valu_insts_per_64_rays": stored.get("valu_insts_per_wave_of_64_rays"),
                "floor_over_executed": (floor / stored["valu_insts_per_wave_of_64_rays"]) if floor and stored.get("valu_insts_per_wave_of_64_rays") else None,
                "hbm_counter_frac": (stored["hbm_bytes_per_launch"] / avg_launch_s / 1e9 / HBM_PEAK_GBS) if stored.get("hbm_bytes_per_launch") else None,
                "csrc_sha16": stored.get("csrc_sha16"),
                "source": "STORED values of rocprofv3 --pmc passes over this command at the same device sources (profiles/counters.json; "
                          "tools/profile_pmc.sh, tools/make_counters.py), not measured in this run",
            }
        if base is not None:
            # The survey's algorithmic-bytes figure: 32 (ray in) + 32 x boxes + 48 x faces the REFERENCE-order walk tests (counted by the
            # oracle on its sample of this frame) + 16 (hit out) + 16 (film) + 16 x P(hit) (texels).  Reported, NOT a bound: the beam
            # kernel walks the tree once per pixel, and the scene is cache-resident, so these bytes never reach HBM.
            bytes_per_ray = 32.0 + 32.0 * base["boxes_per_ray"] + 48.0 * base["faces_per_ray"] + 16.0 + 16.0 + 16.0 * hit_fraction
            alg = bytes_per_ray * my_rays / max(my_secs, 1e-12) / 1e9
            result["hbm_algorithmic"] = {
                "not_a_bound": True, "bytes_per_ray": bytes_per_ray, "boxes_per_ray": base["boxes_per_ray"], "faces_per_ray": base["faces_per_ray"],
                "GBps_if_every_byte_came_from_hbm": alg, "over_hbm_peak": alg / HBM_PEAK_GBS,
            }
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
