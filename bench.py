#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the wavefront path tracer on BASELINE.json's configs[1]
(1 048 576-triangle BVH8 mesh, 1920x1080, pathLength 8, diffuse + conductor), 1..8 MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --config 5        # configs[4]: the 9.9 M-triangle interior whose traversal streams from HBM (1 GPU)
    python bench.py --config 1        # configs[0]: the reference's Cornell box, 512x512, 4 bounces, + the CPU BVH2 path

A "step" is one frame = one primary sample per pixel through the whole hot path (generate, trace, then pathLength x
(the logic step + shade per material type + NEE — one material launch since round 5 —, trace || shadow trace), accumulate).  Frames are rendered in passes (one hipGraph
replay per pass; each frame keeps its own frame number and RNG streams and the result is bit-identical to rendering them
one by one); exactly K frames are timed.  plan_schedule() chooses the pass size from measurements: a budget that fits the
queues (up to 128 full frames' worth of paths) is ONE pass, a longer one is a sequence of --frames-per-pass (64) frame
passes, four in flight.  The K-frame region is timed --reps times, each bracketed by a barrier + device synchronisation,
and the MEDIAN is reported (`steps` stays K, `ms_per_step` = median / K; every repetition is listed in `config.rep_ms`):
at the driver's K = 20 a region is one 24 ms graph replay, and single replays vary by several per cent.  Phase stamps go
to stderr, and a watchdog dumps the Python stacks if a phase takes longer than NX_BENCH_WATCHDOG (150) seconds.  Msamples/s is the reference viewer's
"Megarays/sec": width * height * frames / seconds / 1e6 (/root/reference/Nexus/src/Renderer/Panels/MetricsPanel.cpp:28,35,56).
Scene and BVH live in HBM before the timed region starts.  N > 1: the frame is cut into interleaved 5-row tiles, every
rank renders AND accumulates its tiles with the scene replicated, and the accumulated tiles (16 B per pixel) are gathered
to rank 0 with ONE RCCL gather per pass, where they are scattered into the full image and tonemapped; the image is
bit-identical to the 1-GPU one.  A step on N GPUs is one frame PER GPU (--scaling weak, the default: every GPU keeps the 1-GPU run's
work, K steps = K x N frames, all counted in `value`) or one frame in total (--scaling strong); a weak run on N > 1 GPUs reports the
strong region too (config.strong_scaling).

Rank 0 prints ONE JSON line.  It also carries
  roofline     : the closest-hit trace kernel against the three ceilings that could bind it, all from live launch durations
                 (hipEvents recorded by event nodes around every kernel node of the production hipGraph; every repetition of the
                 timed region is replayed on its own frame numbers and the block is computed from the MEDIAN repetition's
                 replay: roofline.timing lists the launch times of all of them):
                   hbm        : memory-side bytes per launch (`traffic`) = this run's rays per launch x the bytes per ray
                                that `rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE` measured for this workload (committed under
                                profiles/, reads doubled per the gfx950 note of MI355X_MICROARCH.md) / duration / 8 TB/s.
                                The ALGORITHMIC bytes (44 B/ray + 80 B/node visit + 40 B/triangle
                                test + 104 B/instance entry, visits counted by the kernel's counting variant on the same
                                frames) are reported beside it: where the scene fits the caches they are served on-die
                                and exceed the memory-side figure several times.
                   valu-issue : VALU wave-instructions per launch (rays x the profiled instructions per ray) x 2 cycles
                                (a wave64 VALU instruction on a SIMD-32 with several waves resident, MI355X_MICROARCH.md)
                                / (1024 SIMDs x the clock the profile measured) / duration.
                   l1-gather  : 16-byte per-lane loads per launch (rays x TCP_TOTAL_CACHE_ACCESSES per ray of the profile)
                                / duration / 256 CUs against the 3.8 per ns and CU that tools/micro/gather.hip measures for
                                dependent per-lane record gathers that all hit (the traversal's access pattern; 2.4 when they
                                come from the Infinity Cache, 1.0 from HBM).  VALU issue and the gather share the kernel's
                                time nearly additively: neither fraction alone can approach 1.
                 `bound` names the ceiling with the larger fraction; `frac`, `achieved`, `peak`, `unit` belong to it.
  roofline.classes : the second kernel class — the material launch (and, in the classic pipeline, the logic kernel) — per queue
                 item: HBM-side bytes and VALU instructions from the same committed counter passes, against 8 TB/s and the issue rate.
  cpu_baseline : the CPU oracle (port of the reference algorithm) on full frames of the same scene, timed on the
                 host cores of this box (trace, logic and shade threaded).  A reported baseline, not a target.
  emulated_rank (--emulate-rank-of N): rank 0's share of an N-way tile split rendered in this process, against full / N.
Counter constants carry the hash of the kernel sources they were measured on; a mismatch reports `traffic: null` and why.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nexus_amd import capi, multigpu, pod, workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
L2_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md: aggregate L2 bandwidth
SIMDS = 256 * 4         # 256 CUs x 4 SIMD-32
VALU_ISSUE_CYCLES = 2   # wave64 VALU instruction, several waves resident (MI355X_MICROARCH.md, execution model)
# What a wave64 VALU instruction costs a gfx950 SIMD, measured (tools/micro/valu_classes.hip -> profiles/r04_valu_opcode_classes.txt: eight
# waves per SIMD, the one occupancy at which the dispatcher cannot place waves unevenly, and the shader clock read in the kernel):
# ISSUE = one instruction per 2.2 cycles whatever its kind; fp32 fma / mul / add, and / or / xor / bitop3, integer add / sub, right
# shifts and mov need nothing else.  Conversions, min / max, compares, v_cndmask, left shifts, bfe / bfi / perm and the three-operand
# integer forms also occupy a second unit for 4.1 cycles (reciprocals 8.1) that works beside the issue slot in synthetic streams
# (4 fma + 4 cvt alternating: 2.2 per instruction).  (The 2.7 / 4.3 of profiles/r04_valu_cost_microbench.txt, 5 workgroups per CU and a
# nominal 2.4 GHz, were the dispatcher's uneven placement: tools/micro/valu_clock.hip -> r04_valu_issue_by_occupancy.txt.)  The trace
# kernel answers to its instruction COUNT, not to the slow unit: rewriting 12 selects and 8 compare + select pairs per node as
# full-rate instructions (+26 instructions, -34 on the slow unit) made it 1-3 % slower without spills (r04_decode_variants.txt), adding
# 64 FMAs per iteration 15 % slower (NX_EXTRA_VALU).  So the measured issue price is reported beside the guide's 2 cycles.
VALU_CYCLES_MEASURED_ISSUE = 2.2
L1_GATHER_PEAK = 3.8    # 16-byte per-lane loads a CU's vector L1 serves per ns when all of them hit (tools/micro/gather.hip)
CUS = 256
TILE_ROWS = 5           # 1080 = 5 * 216: divides evenly over 1, 2, 4, 8 ranks
COUNTERS_JSON = os.path.join(ROOT, "profiles", "trace_counters.json")


def build_workload(args):
    t0 = time.time()
    if args.config == 1:
        sc = workloads.config1(os.path.join(ROOT, "tests", "golden", "cornell_box.glb"), args.width, args.height, args.path_length)
        name = ("configs[0]: the reference's cornell_box.glb (%d triangles in %d BLASes / instances, BVH8 %d nodes, read by the product's glTF reader), all materials DIFFUSE "
                "with the file's base colours, light emissive (1,1,1) x 35, %dx%d, pathLength %d, MIS on, black background, frames 1..K"
                % (sc.triangles, len(sc.instances), sc.bvh8_nodes, args.width, args.height, args.path_length))
    elif args.config == 2:
        sc = workloads.config2(args.width, args.height, args.nu, args.nv, args.path_length)
        name = ("configs[1]: seeded displaced torus %d triangles (BVH8 %d nodes) + floor + quad light, %dx%d, pathLength %d, conductor(extended)+diffuse, MIS/NEE on"
                % (sc.triangles, sc.bvh8_nodes, args.width, args.height, args.path_length))
    elif args.config == 4:
        sc = workloads.config4(args.width, args.height, args.path_length, split_levels=int(os.environ.get("NX_PROBE_SPLIT_LEVELS", "0")))  # (probe: workloads.split_by_octants)
        sc.env_sampling = True  # configs[3] asks for "HDR envmap NEE/MIS": the extension of nxhip_set_env_sampling
        name = ("configs[3]: %d instances of a %d-triangle BLAS (BVH8 %d nodes) on a jittered lattice, random rotations / scales, DIELECTRIC roughness 0.2 ior 1.45, "
                "2048x1024 procedural environment with NEE / MIS importance sampling (extension), %dx%d, pathLength %d"
                % (len(sc.instances), sc.unique_triangles, sc.bvh8_nodes, args.width, args.height, args.path_length))
    elif args.config == 5:
        sc = workloads.config5(args.width, args.height, args.path_length)
        name = ("configs[4] on 1 GPU: displaced room shell + instanced props, %d triangles in the TLAS (%d unique, BVH8 %d nodes, %.0f MB of nodes + "
                "intersection records: beyond the 256 MiB Infinity Cache), all four material types, textured emissive panels, %dx%d, pathLength %d"
                % (sc.triangles, sc.unique_triangles, sc.bvh8_nodes, sc.scene_bytes() / 1e6, args.width, args.height, args.path_length))
    else:
        raise SystemExit("--config must be 1, 2, 4 or 5")
    return sc, name, time.time() - t0


def upload(ctx, sc, device_bvh=False, device_tlas=False):
    sc.upload(ctx, device_bvh=device_bvh, device_tlas=device_tlas)
    ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)


def tile_pixel_map(width, height, rank, world):
    return multigpu.tile_pixel_map(width, height, rank, world, multigpu.tile_rows_for(height, world, (TILE_ROWS, 8, 4, 6, 3, 2, 1)))


KERNEL_SOURCES = {"trace": ("nx_trace.hip", "nx_traverse.h", "nx_entry.hip", "nx_queue.h", "nx_device.h", "nx_math.h"),
                  "wavefront": ("nx_wavefront.hip", "nx_bsdf.h", "nx_rng.h", "nx_texture.h", "nx_traverse.h", "nx_queue.h", "nx_device.h", "nx_math.h")}


def kernel_source_hash(which):
    """sha256 (16 hex digits) over the device sources a kernel class is compiled from: the committed per-ray / per-item counter
    constants (profiles/trace_counters.json, tools/pmc_to_json.py) carry the hash of the code they were measured on, and a
    figure derived from them is only reported while the code is still that code."""
    import hashlib

    h = hashlib.sha256()
    for f in KERNEL_SOURCES[which]:
        h.update(open(os.path.join(ROOT, "nexus_amd", "csrc", "device", f), "rb").read())
    return h.hexdigest()[:16]


def counters_for(config, key, which):
    """the committed counter constants of one kernel class, or (None, reason)"""
    if not os.path.exists(COUNTERS_JSON):
        return None, "no profiles/trace_counters.json"
    ck = json.load(open(COUNTERS_JSON)).get("config%d" % config, {}).get(key)
    if not ck:
        return None, "no counter pass committed for this workload and kernel class (tools/profile_round.sh)"
    if ck.get("source_sha16") != kernel_source_hash(which):
        return None, "the committed counters were measured on other kernel code (source hash %s, now %s): re-run tools/profile_round.sh" % (ck.get("source_sha16"), kernel_source_hash(which))
    return ck, None


def trace_algorithmic_bytes(st):
    """SURVEY.md section 8d, closest-hit kernel: 44 B per ray (24 in + 20 out), 80 B per node visited, 40 B per triangle
    tested (4 index + 36 positions), 104 B per instance entered."""
    return 44 * st["rays"] + 80 * st["nodes"] + 40 * st["tris"] + 104 * st["instances"]


def cpu_bvh2_baseline(sc, width, height, threads, passes):
    """configs[0]'s "CPU BVH2 intersect reference path" (SURVEY.md section 8c / 8d): the scene flattened to world-space triangles, the
    reference's binned-SAH BVH2 over them (Geometry/BVH/BVH.h:45-65, BVH.cpp:65-210) and the ordered two-child descent of its
    un-included Cuda/BVH/BVH2Traversal.cuh:7-52, as the oracle restates them, on the primary rays of the view: build time and
    Mrays/s on all host threads of the box and on one."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from tests import oracle_lib as O  # test infrastructure, used here only as the reported CPU baseline

    world = workloads.world_triangles(sc)
    rays = workloads.pixel_centre_rays(sc.camera, width, height)
    t0 = time.time()
    b = O._Bvh2()
    assert O.lib().orc_bvh2_build(O._ptr(world), len(world), C.byref(b)) == 0
    t_build = time.time() - t0
    hits = np.zeros(len(rays), dtype=pod.HIT_DT)

    fn, bref, wptr = O.lib().orc_bvh2_trace_closest, C.byref(b), O._ptr(world)
    ray_size, hit_size = rays.dtype.itemsize, hits.dtype.itemsize

    def run(lo, hi):  # one thread's share of the rays, `passes` times over (ctypes releases the GIL during the call)
        for _ in range(passes):
            fn(bref, wptr, rays.ctypes.data + lo * ray_size, hi - lo, hits.ctypes.data + lo * hit_size)

    def timed(nthreads):
        cuts = np.linspace(0, len(rays), nthreads + 1).astype(int)
        t0 = time.time()
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(lambda k: run(int(cuts[k]), int(cuts[k + 1])), range(nthreads)))
        return time.time() - t0

    dt_all, dt_one = timed(threads), timed(1)
    nodes = int(b.nodeCount)
    O.lib().orc_bvh2_free(C.byref(b))
    return {"what": "BVH2 over the %d world-space triangles of the scene, closest hit of the %d pixel-centre primary rays, %d passes" % (len(world), len(rays), passes),
            "build_ms": round(t_build * 1e3, 3), "nodes": nodes,
            "Mrays_per_s": round(len(rays) * passes / dt_all / 1e6, 3), "cores": threads,
            "single_core_Mrays_per_s": round(len(rays) * passes / dt_one / 1e6, 3), "hit_fraction": round(float((hits["hitDistance"] < 1e29).mean()), 4)}


def cpu_baseline(sc, width, height, threads, frames, single_core=True):
    from tests import oracle_lib as O  # test infrastructure, used here only as the reported CPU baseline

    scene = O.OracleScene(sc.blas, sc.instances, sc.tlas_nodes, sc.tlas_idx, sc.materials, sc.lights, sc.camera, sc.settings,
                          sc.diffuse_maps, sc.emissive_maps, sc.hdr_map)
    w = O.Wavefront(scene, width * height, None, pod.RNG_PIXEL_KEYED, pod.CONDUCTOR_EXTENDED)
    t0 = time.time()
    for f in range(1, frames + 1):
        w.render(f, threads=threads)
        w.accumulate(f)
    dt = time.time() - t0
    out = dict(value=width * height * frames / dt / 1e6, unit="Msamples/s", cores=threads, kind="port",
               sample="%d full frame%s (%dx%d, same scene/camera/settings, frames 1..%d); trace, logic and shade on %d pthreads (slots handed out in item order afterwards: "
                      "bit-identical to the serial run), generate / accumulate serial; %.1f s" % (frames, "s" if frames > 1 else "", width, height, frames, threads, dt))
    if single_core:
        # one more frame on a single thread: the 1-core figure SURVEY.md section 8d asks for beside the all-cores one
        t0 = time.time()
        w.render(frames + 1, threads=1)
        dt1 = time.time() - t0
        out.update(single_core_value=round(width * height / dt1 / 1e6, 4), single_core_sample="1 full frame on 1 thread; %.1f s" % dt1)
    return out


_T0 = time.time()
_WATCHDOG_S = float(os.environ.get("NX_BENCH_WATCHDOG", "150"))


def stamp(phase):
    """One line per phase on stderr, and the watchdog re-armed for the next one.  A run that stops making progress (a hung
    device call, a runtime that never comes up) then leaves the name of the last phase it reached behind, and after
    NX_BENCH_WATCHDOG seconds (default 150: below the 200-300 s `timeout` of the sweep scripts and of one bench run inside a
    gpurun call) every thread's Python stack, instead of dying silently with both streams empty."""
    import faulthandler

    sys.stderr.write("[bench %7.2f s] %s\n" % (time.time() - _T0, phase))
    sys.stderr.flush()
    faulthandler.dump_traceback_later(_WATCHDOG_S, exit=True)


def plan_schedule(steps, cap, passes_in_flight, explicit_pass_size, share, pixels=1920 * 1080):
    """Pass size S and passes in flight R for a timed region of `steps` frames.  `cap`: the pass size of a long run (frames),
    `share`: the fraction of the image this process renders (1 / ranks).  Measured (gpurun_out r3_15 / r3_16, DESIGN.md section
    6): a region that starts and ends with an idle GPU is fastest as ONE pass, whatever its length — 20 / 40 / 64 / 128 frames:
    1 654 / 1 835 / 1 964 / 1 963 Msamples/s as one pass against 1 622 / 1 803 / 1 877 / 1 907 for the best split into passes in
    flight (passes that start together reach their drains together, and every pass pays its own nine trace levels' worth of
    slowest rays) — as long as its paths fit the queues (128 full frames' worth: 72 GB).  A longer budget is a sequence of
    `cap`-frame passes, four in flight: those do run out of phase, and the drain of one is filled by the bulk of the next
    (512 frames: 1 990 against 1 905 with one at a time)."""
    single = int(128 * (1920 * 1080) / (pixels * share))  # frames whose paths fill the queues of 128 full 1080p frames
    if explicit_pass_size:
        S = max(1, min(cap, steps))
    else:
        S = steps if steps <= single else cap
    n_passes = -(-steps // S)
    small = S * share <= 4.0
    R = passes_in_flight if passes_in_flight else (6 if small else 4)
    return S, max(1, min(R, 8, n_passes)), n_passes


def main():
    stamp("start")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="frames timed per repetition (default 512; --config 5: 64)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames first (default 64; --config 5: 16)")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-frame timed region; the median is reported")
    ap.add_argument("--config", type=int, default=2, help="1 = configs[0] (the reference's Cornell box, 512x512, pathLength 4, diffuse only; the CPU baseline adds the BVH2 path); 2 = BASELINE.json configs[1] (the metric's workload); 4 = configs[3] (instanced TLAS, dielectric, environment NEE / MIS); 5 = configs[4] on one GPU (HBM-resident scene)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--path-length", type=int, default=None)
    ap.add_argument("--nu", type=int, default=1024, help="config 2 torus grid: 2*nu*nv triangles")
    ap.add_argument("--nv", type=int, default=512)
    ap.add_argument("--frames-per-pass", type=int, default=None,
                    help="frames batched into one wavefront pass on 1 GPU (default 64; --config 5: 16; N ranks: N times as many, at most 512, so that a "
                         "rank's pass keeps its size); a step budget that is not a multiple ends with one shorter pass")
    ap.add_argument("--passes-in-flight", type=int, default=None,
                    help="passes rendered concurrently on separate streams (nxhip_set_passes_in_flight): the drain of one pass overlaps the bulk of the next. "
                         "Default: 6 for passes of up to 4 frames, else 4; never more than the timed region has passes")
    ap.add_argument("--pixel-order", choices=["rows", "tiles"], default="tiles", help="order of a rank's paths: image rows, or 8x8 tiles")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="what a step is on N GPUs.  weak (default): one frame PER GPU — a step renders N frames, each tile-split over the N GPUs, so every GPU keeps "
                         "the work of the 1-GPU run (K steps = K x N frames; Msamples/s counts them all).  strong: one frame in total, tile-split — K steps = K frames "
                         "whatever N (a rank of 8 then renders 2.5 frames' worth of paths per 20-step region: the latency floor of a pass, DESIGN.md section 5). "
                         "Identical at N = 1.  A weak run at N > 1 also times the strong region and reports it as config.strong_scaling")
    ap.add_argument("--host-bvh", action="store_true", help="upload the BLASes of the host builder (binned SAH with 8 bins + SAH-DP collapse on CPU threads: the reference's "
                                                            "algorithm) instead of building them on the GPU (nxhip_build_blas: the same rule top-down on the device with 16 bins, "
                                                            "then the same collapse; 36 ms instead of 0.4 s per million triangles and 2 %% fewer node visits per ray)")
    ap.add_argument("--device-bvh", action="store_true", help="(the default since round 3; kept so that older command lines still parse)")
    ap.add_argument("--host-tlas", action="store_true", help="with the device-built BLASes: the TLAS from the host builder (the reference's agglomerative clustering + collapse) "
                                                             "instead of nxhip_rebuild_tlas")
    ap.add_argument("--pass-sizes", type=str, default="", help="experiment: explicit pass sizes of the timed region, e.g. 8,6,4,2 (must sum to --steps)")
    ap.add_argument("--emulate-rank-of", type=int, default=0,
                    help="N: after the full-frame measurement, render rank 0's share of an N-way tile split (same tiles, pass sizes and passes in flight as a "
                         "rank of bench.py --gpus N, no collective) in this process and report per_rank_ms beside full_ms / N: the communication-free "
                         "scaling efficiency a 1-GPU box can measure")
    ap.add_argument("--no-obj-check", action="store_true", help="config 2: skip the .obj round trip.  By default the mesh is written as a Wavefront .obj, read back with the "
                                                                "product's OBJLoader and required to equal the in-memory triangles the BVH was built from (untimed, ~16 s; "
                                                                "configs[1] says 'single 1M-triangle .obj mesh')")
    ap.add_argument("--no-reference-mode", action="store_true",
                    help="skip the second measurement: the same timed region with the REFERENCE's semantics — random numbers keyed by queue slot, slots handed out in "
                         "serial order (grid-wide ordered compaction), no conductor kernel — reported as config.reference_mode beside the headline")
    ap.add_argument("--no-entry-points", action="store_true",
                    help="start every primary ray at the TLAS root, as the reference does, instead of at the state the first node steps of its run of 64 paths provably "
                         "share (nxhip_set_entry_points; hit records are the same bit for bit either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--png", type=str, default="", help="write the accumulated image here (rank 0)")
    args = ap.parse_args()
    big = args.config == 5
    cornell = args.config == 1
    args.width = args.width or (3840 if big else 512 if cornell else 1920)
    args.height = args.height or (2160 if big else 512 if cornell else 1080)
    args.path_length = args.path_length or (16 if big else 4 if cornell else 8)
    args.steps = args.steps if args.steps is not None else (64 if big else 512)
    args.warmup = args.warmup if args.warmup is not None else (16 if big else 64)

    # stdout carries exactly one line, the JSON: libraries that print banners to stdout (RCCL does at init) are sent to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched through torch.distributed.run with N ranks")
        args.gpus = world

    # NX_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, gather, compose) with whatever world size the launcher
    # gave, even 1 — a rehearsal of the RCCL path on a single-GPU box
    dist_mode = world > 1 or bool(os.environ.get("NX_BENCH_FORCE_DIST"))
    if dist_mode:
        # (this pool's host driver only supports dmabuf IPC: without the setting RCCL's peer buffers fail with
        #  "hipIpcGetMemHandle: invalid argument"; the launcher exports it, a bare `torchrun bench.py` may not)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # torch ships its own libamdhip64 / libhsa-runtime64: import it BEFORE libnexus_amd.so is loaded so that the
        # library's libamdhip64.so.7 dependency binds to the copy torch already holds (two HIP runtimes in one process
        # cannot share the device, nor a stream handle).
        import torch  # noqa: F401
        import torch.distributed  # noqa: F401

    W, H = args.width, args.height
    if args.steps < 1 or args.warmup < 0 or args.reps < 1:
        raise SystemExit("--steps and --reps must be >= 1 and --warmup >= 0")
    # frames per step: one per GPU under weak scaling (per-GPU work fixed as N grows), one in total under strong scaling
    F = world if args.scaling == "weak" else 1
    n_frames, n_warm = args.steps * F, args.warmup * F
    explicit_fpp = args.frames_per_pass is not None
    cap = max(1, min((args.frames_per_pass or (16 if big else 64)) * world, 512))
    if args.config == 4 and not args.passes_in_flight:
        args.passes_in_flight = 1  # configs[3] (heavily instanced: long trace launches, little to overlap): 271 Msamples/s with one pass at a time, 254 with four
    S, R, n_passes = plan_schedule(n_frames, cap, args.passes_in_flight, explicit_fpp, 1.0 / world, args.width * args.height)
    pass_sizes = [int(x) for x in args.pass_sizes.split(",")] if args.pass_sizes else None
    if pass_sizes:
        if sum(pass_sizes) != n_frames or min(pass_sizes) < 1:
            raise SystemExit("--pass-sizes must be positive and sum to --steps (x the number of GPUs under weak scaling)")
        S = max(pass_sizes)
        R = max(1, min(args.passes_in_flight or R, 8, len(pass_sizes)))
    if R > 1 or args.emulate_rank_of > 1:
        # concurrent passes need a hardware queue per stream and graph branch; the HIP runtime's default of 4 serialises them.
        # Must be in the environment before the process first touches HIP (nothing has yet).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

    def schedule(frames, size=None):
        """pass sizes that render exactly `frames` frames"""
        size = size or S
        if pass_sizes and frames == n_frames and size == S:
            return list(pass_sizes)
        return [size] * (frames // size) + ([frames % size] if frames % size else [])

    sc, workload_name, t_build = build_workload(args)
    stamp("scene built on the host (%.1f s)" % t_build)
    obj_check = None
    if args.config == 2 and not args.no_obj_check and rank == 0:
        # (rank 0 only: eight ranks writing and reading a 120 MB file at once is 8 x 3 s of host work and file I/O in front of the
        #  first barrier for a check whose answer does not depend on the rank)
        obj_check = workloads.check_obj_round_trip(sc)
        stamp("mesh written as .obj, read back by OBJLoader and compared")
    if os.environ.get("NX_BENCH_NO_MIS"):  # experiment only: how much of the shade kernels is next-event estimation
        sc.settings["useMIS"] = 0
        workload_name += " [useMIS off: experiment]"

    if os.environ.get("NX_BENCH_NO_ENV_SAMPLING"):  # experiment only: the share of environment importance sampling
        sc.env_sampling = False
        workload_name += " [environment sampling off: experiment]"

    frames_rendered = [0]  # by step(): what a profiler run of this command has seen of every kernel (tools/pmc_to_json.py)
    dist = None
    torch = None
    if dist_mode:
        import torch
        import torch.distributed as dist

        # NX_BENCH_BACKEND=gloo + NX_BENCH_SHARE_GPU=1 rehearse the N > 1 path on a single GPU (tiles staged through
        # host memory); the real run is one rank per GPU over RCCL ("nccl").
        backend = os.environ.get("NX_BENCH_BACKEND", "nccl")
        dev = 0 if os.environ.get("NX_BENCH_SHARE_GPU") else local_rank
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend)
        # One explicit (non-default) torch stream carries everything: the context launches its kernels on it, torch ops
        # and the collective are ordered on it, so the path needs no host synchronisation between render, gather and
        # accumulate.  (The default stream's handle is 0, which nxhip_create reads as "make your own stream".)
        side = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(side)
        assert side.cuda_stream != 0
        ctx = capi.Context(W, H, device=dev, stream=side.cuda_stream)
    else:
        ctx = capi.Context(W, H, device=0)
    stamp("device context created")
    args.device_bvh = not args.host_bvh and not cornell  # (configs[0]: the host builders — the reference's algorithm, the bytes the oracle's builders produce)
    upload(ctx, sc, device_bvh=args.device_bvh, device_tlas=args.device_bvh and not args.host_tlas)
    if not args.no_entry_points:
        ctx.set_entry_points(True)
    stamp("scene uploaded")

    if dist_mode:
        pm = tile_pixel_map(W, H, rank, world)
        if args.pixel_order == "tiles":
            pm = multigpu.tiled_order(pm, W)
        ctx.set_pixel_map(pm)
        ctx.set_frames_per_pass(S)
        if R > 1:
            ctx.set_passes_in_flight(R)
        n_local = len(pm)

        class _DeviceArray:  # zero-copy torch view of the context's accumulation tile (float4 per local pixel)
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}

        acc_tile = torch.as_tensor(_DeviceArray(ctx.accumulation_device_ptr(), (n_local, 4)), device="cuda")
        assert acc_tile.data_ptr() == int(ctx.accumulation_device_ptr())
        if rank == 0:
            gathered = [torch.zeros((n_local, 4), dtype=torch.float32, device="cuda") for _ in range(world)]
            full_acc = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
            full_rgba = torch.zeros((W * H,), dtype=torch.int32, device="cuda")
            maps_dev = []
            seen = np.zeros(W * H, dtype=bool)
            for r in range(world):
                m = tile_pixel_map(W, H, r, world)
                if args.pixel_order == "tiles":
                    m = multigpu.tiled_order(m, W)
                assert len(m) == n_local and not seen[m].any()
                seen[m] = True
                maps_dev.append(torch.from_numpy(m.astype(np.int64)).to("cuda").to(torch.int32))
            assert seen.all()

        def step(n):
            frames_rendered[0] += n
            # every rank path-traces and accumulates its own tiles (the exact per-frame running mean of the 1-GPU path) ...
            if ctx.frames_per_pass != n:
                ctx.set_frames_per_pass(n)  # within the allocated capacity: no synchronisation, the size travels as a kernel argument
            ctx.render_frame()
            ctx.accumulate()
            # ... and the one collective of the path moves the accumulated tiles (16 B per pixel, once per pass) to rank 0
            # over xGMI, which scatters them into the full image and tonemaps
            if backend == "nccl":
                dist.gather(acc_tile, gathered if rank == 0 else None, dst=0)
            else:
                host = acc_tile.cpu()
                parts = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0:
                    for r in range(world):
                        gathered[r].copy_(parts[r])
            if rank == 0:
                for r in range(world):
                    ctx.compose_tiles(gathered[r].data_ptr(), n_local, maps_dev[r].data_ptr(), full_acc.data_ptr(), full_rgba.data_ptr())

        def sync():
            torch.cuda.synchronize()
            t_local = time.perf_counter()  # this rank's own work (rank 0: including the tiles it waited for) is done
            dist.barrier()
            torch.cuda.synchronize()
            return t_local
    else:
        if args.pixel_order == "tiles":
            ctx.set_pixel_order(pod.ORDER_TILES)  # (the library's own 8 x 8 tile order: what a C++ caller gets from PathTracer::SetPixelOrder)
        ctx.set_frames_per_pass(S)
        if R > 1:
            ctx.set_passes_in_flight(R)

        def step(n):
            frames_rendered[0] += n
            if ctx.frames_per_pass != n:
                ctx.set_frames_per_pass(n)  # within the allocated capacity: no synchronisation, the size travels as a kernel argument
            ctx.render_frame()
            ctx.accumulate()

        def sync():
            ctx.sync()
            return time.perf_counter()

    per_rank_ms = {}  # label -> per repetition, every rank's own time to the end of its work (before the closing barrier)

    def measure(steps, warmup, reps, size, label):
        """`reps` timed regions of exactly `steps` frames (pass sizes: schedule(steps, size)), after `warmup` untimed frames in
        passes of the same sizes — so that every slot has instantiated the graph the timed passes replay (a pass of another size
        class would be a different graph: nxhip keeps one instance per shape) and touched its queues."""
        # every slot once at the timed pass size (one slot: the context itself), then the image starts over
        for n in schedule(steps, size)[:1] * max(R, 1):
            step(n)
        sync()
        ctx.reset_frame_number()
        for n in schedule(warmup, size):
            step(n)
        sync()
        stamp("%s: warm-up done (%d frames)" % (label, warmup))
        out = []
        for k in range(reps):
            sync()
            t0 = time.perf_counter()
            for n in schedule(steps, size):
                step(n)
            t_local = sync()
            dt = time.perf_counter() - t0
            if dist_mode:
                t = torch.tensor([dt], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
                # every rank's own figure goes to rank 0 as well, so that a bad scaling curve can be read from ONE run: which rank
                # was slow, and whether the ranks finished together (load balance of the tile split) or the root waited
                mine = torch.tensor([(t_local - t0) * 1e3], dtype=torch.float64, device="cuda")
                everyone = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(everyone, mine)
                per_rank_ms.setdefault(label, []).append([round(float(x.item()), 3) for x in everyone])
            out.append(dt)
            stamp("%s: repetition %d of %d: %.3f ms" % (label, k + 1, reps, dt * 1e3))
        return out

    # one step = one frame; a pass renders up to S frames
    rep_s = measure(n_frames, n_warm, args.reps, S, "timed region")
    elapsed = statistics.median(rep_s)
    if os.environ.get("NX_WAVE_TIMELINE_OUT"):
        # measurement builds only (tools/build_variant.sh ... -DNX_WAVE_TIMELINE): every trace wave's start / queue-dry / end times of
        # the last pass, [any-hit][bounce][wave][5] uint64, for tools/wave_timeline.py
        import ctypes
        import numpy as _np
        _L = capi.lib()
        if hasattr(_L, "nxhip_debug_read_wave_timeline"):
            _buf = _np.zeros((2, 12, 8192, 5), _np.uint64)
            _rc = _L.nxhip_debug_read_wave_timeline(_buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_ulonglong(_buf.nbytes))
            stamp("wave timeline read: rc %d" % _rc)
            _np.save(os.environ["NX_WAVE_TIMELINE_OUT"], _buf)

    # (--png shows the headline's image, not what later measurements add to the accumulation)
    headline_full = full_rgba.clone() if (dist_mode and args.png and rank == 0) else None

    # a weak-scaling run on N > 1 GPUs also times the strong-scaling region — K frames in total, tile-split — as a second figure
    strong = None
    if dist_mode and world > 1 and F > 1:
        S_strong, _, _ = plan_schedule(args.steps, cap, 1, explicit_fpp, 1.0 / world, W * H)
        ss = measure(args.steps, args.warmup, 3, S_strong, "strong-scaling region")
        med = statistics.median(ss)
        strong = {"what": "the same K steps with ONE frame per step, tile-split over the N GPUs (total work fixed: --scaling strong), median of 3 repetitions",
                  "value": round(W * H * args.steps / med / 1e6, 3), "unit": "Msamples/s", "frames_timed": args.steps, "frames_per_pass": S_strong,
                  "ms_per_step": round(med / args.steps * 1e3, 4), "rep_ms": [round(x * 1e3, 3) for x in ss]}
        sync()
        ctx.set_frames_per_pass(S)

    emulated = None
    if args.emulate_rank_of > 1 and not dist_mode:
        # Rank 0's share of an N-way split, as bench.py --gpus N would give it to a rank: its tiles, the pass plan of a rank,
        # no collective.  per_rank_ms against full_ms / N is the scaling efficiency before communication.
        N = args.emulate_rank_of
        pm = tile_pixel_map(W, H, 0, N)
        if args.pixel_order == "tiles":
            pm = multigpu.tiled_order(pm, W)
        # (an explicit --frames-per-pass is taken as the rank's pass size itself, for sweeps)
        capN = max(1, min(args.frames_per_pass if explicit_fpp else (16 if big else 64) * N, 512))
        S_full, R_full = S, R
        emu_frames = args.steps * (N if args.scaling == "weak" else 1)  # a rank's share: N x K frames of 1 / N of the pixels, or K frames of them
        emu_warm = args.warmup * (N if args.scaling == "weak" else 1)
        S, R, _ = plan_schedule(emu_frames, capN, args.passes_in_flight, explicit_fpp, 1.0 / N, W * H)
        sync()
        ctx.set_pixel_map(pm)
        ctx.set_frames_per_pass(S)
        ctx.set_passes_in_flight(max(R, 1))
        rank_s = measure(emu_frames, emu_warm, args.reps, S, "rank 0 of %d" % N)
        per_rank = statistics.median(rank_s)
        ideal = elapsed * emu_frames / n_frames / N  # the rank's share of the work at the 1-GPU rate
        emulated = {"ranks": N, "scaling": args.scaling, "frames": emu_frames, "local_pixels": int(len(pm)), "frames_per_pass": S, "passes_in_flight": R,
                    "per_rank_ms": round(per_rank * 1e3, 3), "full_ms": round(elapsed * 1e3, 3), "ideal_ms": round(ideal * 1e3, 3),
                    "efficiency_without_communication": round(ideal / per_rank, 4),
                    "msamples_per_s_if_all_ranks_matched": round(W * H * emu_frames / per_rank / 1e6, 1), "rep_ms": [round(x * 1e3, 3) for x in rank_s]}
        # back to the full frame for the roofline section below
        S, R = S_full, R_full
        sync()
        ctx.set_pixel_order(pod.ORDER_TILES if args.pixel_order == "tiles" else pod.ORDER_ROWS)
        ctx.set_frames_per_pass(S)
        ctx.set_passes_in_flight(max(R, 1))

    # ---- the same region with the reference's own semantics (PathTracer.cu:143, 326, 475-478; Random.cuh:79-82): the headline
    # runs pixel-keyed random numbers, racing slot allocation and the extended conductor kernel; these two lines say what the
    # serial slot order costs on the whole chip (`ordered`: same work as the headline) and what the reference's exact
    # configuration renders at (`reference`: conductor hits end unshaded, as the reference's commented-out kernel leaves them)
    reference_mode = None
    headline_rgba8 = ctx.read_rgba8() if (args.png and not dist_mode) else None  # (--png shows the headline's image, not the second measurement's)
    if not args.no_reference_mode and not dist_mode:
        reference_mode = {"what": "the timed region again (median of 3 repetitions) with random numbers keyed by queue slot and slots in the reference's serial order "
                                  "(NX_RNG_REFERENCE_SLOT, NX_COMPACT_ORDERED: tiles by ticket + decoupled look-back on all CUs); `reference` also drops the conductor kernel "
                                  "(NX_CONDUCTOR_REFERENCE), i.e. less work per frame than the headline; `ordered` keeps it (same work)"}
        for key, conductor in (("ordered", pod.CONDUCTOR_EXTENDED), ("reference", pod.CONDUCTOR_REFERENCE)):
            sync()
            ctx.set_modes(pod.RNG_REFERENCE_SLOT, pod.COMPACT_ORDERED, conductor)
            rs = measure(n_frames, min(n_warm, S), 3, S, "%s mode" % key)
            med = statistics.median(rs)
            reference_mode[key] = {"value": round(W * H * n_frames / med / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(med / args.steps * 1e3, 4),
                                   "rep_ms": [round(x * 1e3, 3) for x in rs], "vs_headline": round(elapsed / med, 4)}
        sync()
        ctx.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
        ctx.reset_frame_number()

    value = W * H * n_frames / elapsed / 1e6
    out = {
        "metric": "Msamples/sec (rays traced/sec) at 1080p, 8-bounce, 1M-tri BVH8; 1/2/4/8 GPU",  # BASELINE.json's metric, verbatim
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload_name,
            "parallelism": "1 GPU" if world == 1 else "%d GPUs: interleaved %d-row tiles, scene replicated, per-rank accumulation, one RCCL gather of accumulated tiles per pass" % (world, TILE_ROWS),
            "rng": "pixel-keyed", "compaction": "workgroup-aggregated atomics", "entry_points": not args.no_entry_points, "launch": "one hipGraph replay per pass of up to %d frames" % S,
            "frames_per_pass": S, "passes_in_flight": R, "pixel_order": args.pixel_order, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "step": "one frame" if F == 1 else "%d frames (one per GPU), each tile-split over the %d GPUs" % (F, world),
            "frames_per_step": F, "frames_timed": n_frames, "samples_timed": int(W) * int(H) * n_frames,
            "timing": "median of %d repetitions of the %d-step = %d-frame region, each bracketed by barrier + device sync" % (args.reps, args.steps, n_frames),
            "rep_ms": [round(x * 1e3, 3) for x in rep_s],
            "mean_rep_ms": round(statistics.mean(rep_s) * 1e3, 3), "value_from_the_mean": round(W * H * n_frames / statistics.mean(rep_s) / 1e6, 3),
            "pass_sizes": schedule(n_frames), "frames_rendered_by_the_timed_loop": frames_rendered[0],
            "host_scene_build_s": round(t_build, 2), "tlas_builder": "device (nxhip_rebuild_tlas)" if (args.device_bvh and not args.host_tlas) else "host: agglomerative clustering + SAH-DP collapse (the reference's algorithm)",
            "build": (lambda f: {"gfx950": bool(f & 1), "scheduler_flags": bool(f & 2), "debug_hooks": bool(f & 4)})(int(capi.lib().nxhip_build_info())),
            "blas_builder": "device: top-down binned SAH (16 bins) + SAH-DP collapse, a primitive priced at 0.75 of a node (nxhip_build_blas)" if args.device_bvh else "host: binned SAH (8 bins) + SAH-DP collapse (the reference's algorithm, --host-bvh)",
        },
    }

    if dist_mode and per_rank_ms.get("timed region"):
        rows = per_rank_ms["timed region"]
        med = [round(statistics.median(r[k] for r in rows), 3) for k in range(world)]
        out["config"]["per_rank"] = {
            "what": "each rank's own time for the K-frame region, in ms: from the common start to the end of its own work (render, accumulate, its side of the "
                    "gather; rank 0: including the tiles it waited for and the compose), before the closing barrier; `rep_ms` is the slowest rank + barrier",
            "rep_ms_by_rank": rows, "median_ms_by_rank": med, "slowest_rank": int(max(range(world), key=lambda k: med[k])),
            "spread": round(max(med) / max(1e-9, min(med)), 4),
            "gather_bytes_per_rank_and_pass": int(n_local * 16), "backend": backend}
    if strong:
        out["config"]["strong_scaling"] = strong
    if dist_mode and world > 1:
        # both meanings of a step as top-level keys, so that one SCALE record shows the two curves: weak = one frame per GPU and step
        # (K x N frames, every GPU keeps the 1-GPU run's work), strong = one frame per step in total, tile-split (the definition of
        # rounds 1-3 and of north_star's "frames are tiled across the GPUs")
        out["value_weak"] = out["value"] if args.scaling == "weak" else None
        out["value_strong"] = strong["value"] if strong else (out["value"] if args.scaling == "strong" else None)
        # ... and the same K steps on ONE GPU inside this job (rank 0 alone, full frame, the others wait at a barrier), so that
        # the line carries its own denominator: efficiency = value_N / (N x value_1)
        single = None
        if rank == 0:
            ctx.sync()
            full_map = multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W) if args.pixel_order == "tiles" else None
            S1, _, _ = plan_schedule(args.steps, max(1, min(args.frames_per_pass or (16 if big else 64), 512)), 1, explicit_fpp, 1.0, W * H)
            ctx.set_pixel_map(full_map)
            ctx.set_frames_per_pass(S1)
            ctx.set_passes_in_flight(1)
            sizes1 = [S1] * (args.steps // S1) + ([args.steps % S1] if args.steps % S1 else [])

            def run1(sizes):
                for n in sizes:
                    if ctx.frames_per_pass != n:
                        ctx.set_frames_per_pass(n)
                    ctx.render_frame()
                    ctx.accumulate()
                ctx.sync()

            run1(sizes1[:1])
            ctx.reset_frame_number()
            run1([S1] * (args.warmup // S1) + ([args.warmup % S1] if args.warmup % S1 else []))
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                run1(sizes1)
                ts.append(time.perf_counter() - t0)
            med1 = statistics.median(ts)
            v1 = W * H * args.steps / med1 / 1e6
            single = {"what": "the same K steps rendered by rank 0 alone on its one GPU (full frame, one frame per step, median of 3 repetitions), inside this job",
                      "value": round(v1, 3), "ms_per_step": round(med1 / args.steps * 1e3, 4), "rep_ms": [round(x * 1e3, 3) for x in ts]}
            out["single_gpu_in_this_job"] = single
            out["efficiency_weak"] = round(out["value_weak"] / (world * v1), 4) if out["value_weak"] else None
            out["efficiency_strong"] = round(out["value_strong"] / (world * v1), 4) if out["value_strong"] else None
            # back to rank 0's tiles for the roofline section (the pixel-set changes re-allocated the accumulation: the zero-copy view
            # of it is stale from here on, and nothing below gathers any more)
            acc_tile = None
            ctx.set_pixel_map(pm)
            ctx.set_frames_per_pass(S)
            if R > 1:
                ctx.set_passes_in_flight(R)
        dist.barrier()
    if reference_mode:
        out["config"]["reference_mode"] = reference_mode
    if emulated:
        out["emulated_rank"] = emulated
    if obj_check:
        out["config"]["obj_round_trip"] = obj_check

    # ---- roofline of the dominant kernel (closest-hit trace), on rank 0: at N > 1 these are rank 0's launches on ITS tiles (rays per
    # launch, durations, fractions per GPU); the section runs no collective, the other ranks wait at the closing barrier
    if rank == 0 and not args.no_roofline:
        stamp("roofline: in-graph kernel timing over the frames of the timed repetitions + counting variant on the median one")
        if ctx.frames_per_pass != S:
            ctx.set_frames_per_pass(S)
        rep_sizes = schedule(n_frames)
        warm_sizes = schedule(n_warm, S)

        def run_passes(sizes):
            # (no collective here: at N > 1 the other ranks wait at the closing barrier while rank 0 measures its own launches)
            for n in sizes:
                if ctx.frames_per_pass != n:
                    ctx.set_frames_per_pass(n)
                ctx.render_frame()
                ctx.accumulate()

        # (b) durations: hipEvent pair around every kernel of the production graph (event-record nodes inside the hipGraph, so the
        # closest-hit and shadow traces of a bounce overlap exactly as in the timed region), over THE SAME FRAME NUMBERS as the
        # timed repetitions: the image starts over, the warm-up frames are rendered again, then one series of replays per
        # repetition; the events of a repetition's last replay are read after it.  Which frames a pass renders decides how long
        # it takes (some frame ranges hold rays of thousands of traversal steps: DESIGN.md section 6), so the block below is
        # computed from the repetition whose wall time was the MEDIAN of the timed region — the run `value` comes from.
        order = sorted(range(len(rep_s)), key=lambda k: rep_s[k])
        med_rep = order[len(order) // 2]
        ctx.reset_frame_number()
        run_passes(warm_sizes)
        ctx.sync()
        ctx.enable_kernel_timing(True, in_graph=True, last_replay_only=True)
        ctx.read_kernel_times(reset=True)
        run_passes(rep_sizes[:1])  # (builds the instrumented graph; its frames are rendered again below)
        ctx.sync()
        ctx.read_kernel_times(reset=True)
        ctx.set_frame_number(n_warm)
        timed = []
        for k in range(len(rep_s)):
            t0 = time.perf_counter()
            run_passes(rep_sizes)
            ctx.sync()
            wall = time.perf_counter() - t0
            tl = ctx.read_graph_timeline()
            kt_k = ctx.read_kernel_times(reset=True)
            span = max((s0 + d0) for _, s0, d0 in tl) if tl else 0.0
            timed.append({"wall_ms": wall * 1e3, "kt": kt_k, "timeline": tl, "span_ms": span})
        ctx.enable_kernel_timing(False)
        kt = timed[med_rep]["kt"]
        # (a) units: the counting variant of the same kernel over the frames of that repetition
        ctx.set_frame_number(n_warm + med_rep * n_frames)
        ctx.enable_trace_stats(True)
        ctx.read_trace_stats(reset=True)
        for n in rep_sizes:
            if ctx.frames_per_pass != n:
                ctx.set_frames_per_pass(n)
            ctx.render_frame()
        closest, shadow = ctx.read_trace_stats(reset=True)
        ctx.enable_trace_stats(False)
        q = ctx.read_queue_sizes()  # (of the repetition's last pass)
        passes = len(rep_sizes)
        frames = n_frames
        last_pass_frames = rep_sizes[-1]
        launches = max(1, kt["trace"]["launches"])        # closest-hit launches of one replay (pathLength + 1)
        avg_ms = kt["trace"]["ms"] / launches
        dur_s = max(avg_ms * 1e-3, 1e-12)
        rays_per_launch = closest["rays"] / (launches * passes)                        # the counting run covered `passes` replays
        alg_bytes_per_launch = trace_algorithmic_bytes(closest) / (launches * passes)
        alg_gbs = alg_bytes_per_launch / dur_s / 1e9
        frames_timed = {k: (n_frames if k == "accumulate" else last_pass_frames) for k in kt}  # the instrumented replay is the repetition's last pass; accumulate is launched outside the graph, once per pass
        # Counter-side constants of this workload, per ray, from the committed rocprofv3 --pmc passes (profiles/README.md):
        # HBM-side bytes (FETCH_SIZE x 2 + WRITE_SIZE), VALU wave-instructions and the clock under the profiler.  Per ray they
        # do not depend on the pass size (measured equal within 4 % at 1 and 20 frames per pass), so the driver's pass size
        # gets the same figures as the profiled one.
        ck, ck_reason = counters_for(args.config, "trace_closest", "trace")
        ceilings = {}
        traffic = None
        if ck:
            traffic = int(rays_per_launch * (ck["hbm_read_bytes_per_ray"] + ck["hbm_write_bytes_per_ray"]))
            hbm_gbs = traffic / dur_s / 1e9
            ceilings["hbm"] = {"achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                               "bytes_per_ray": round(ck["hbm_read_bytes_per_ray"] + ck["hbm_write_bytes_per_ray"], 1)}
            clock = ck["clock_GHz"]
            ginst = rays_per_launch * ck["valu_insts_per_ray"] / dur_s / 1e9
            peak_ginst = SIMDS * clock / VALU_ISSUE_CYCLES
            ceilings["valu-issue"] = {"achieved": round(ginst, 1), "peak": round(peak_ginst, 1), "unit": "G wave-instructions/s", "frac": round(ginst / peak_ginst, 4),
                                      "valu_insts_per_ray": round(ck["valu_insts_per_ray"], 1), "cycles_per_instruction": VALU_ISSUE_CYCLES, "clock_GHz": clock}
            # The any-hit launch of a bounce runs concurrently with the closest-hit one on the same SIMDs, and a wave64 VALU
            # instruction measures 2.2 cycles, not 2 (above): the SIMDs' issue slots are fuller than `frac` says.  Reported
            # beside it, not as the fraction.
            cks, _ = counters_for(args.config, "trace_shadow", "trace")
            if cks and kt["shadow"]["launches"]:
                shadow_rays_per_launch = shadow["rays"] / (kt["shadow"]["launches"] * passes)
                both = ginst + shadow_rays_per_launch * cks["valu_insts_per_ray"] / dur_s / 1e9
                ceilings["valu-issue"]["frac_with_concurrent_any_hit"] = round(both / peak_ginst, 4)
                ceilings["valu-issue"]["frac_with_concurrent_any_hit_at_the_measured_%.1f_cycles" % VALU_CYCLES_MEASURED_ISSUE] = round(both / peak_ginst * VALU_CYCLES_MEASURED_ISSUE / VALU_ISSUE_CYCLES, 4)
            ceilings["valu-issue"]["frac_at_the_measured_%.1f_cycles" % VALU_CYCLES_MEASURED_ISSUE] = round(ginst / peak_ginst * VALU_CYCLES_MEASURED_ISSUE / VALU_ISSUE_CYCLES, 4)
            if "l1_accesses_per_ray" in ck:
                loads = rays_per_launch * ck["l1_accesses_per_ray"] / (dur_s * 1e9) / CUS
                ceilings["l1-gather"] = {"achieved": round(loads, 3), "peak": L1_GATHER_PEAK, "unit": "16-B lane-loads/ns/CU", "frac": round(loads / L1_GATHER_PEAK, 4),
                                         "lane_loads_per_ray": round(ck["l1_accesses_per_ray"], 1)}
        bound = max(ceilings, key=lambda k: ceilings[k]["frac"]) if ceilings else "hbm"
        top = ceilings.get(bound, {"achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None})
        out["roofline"] = {
            "scope": "the whole frame on the one GPU" if world == 1 else "rank 0 of %d: its %d of %d pixels (every rank runs the same kernels on an equal share)" % (world, n_local, W * H),
            "bound": bound, "kernel": "trace_kernel<closest>", "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
            "traffic": traffic,
            "traffic_source": ("rays per launch of this run x bytes per ray from %s (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; reads x2 per the gfx950 note of "
                               "MI355X_MICROARCH.md)" % os.path.relpath(COUNTERS_JSON, ROOT)) if ck else ck_reason,
            "ceilings": ceilings,
            # the algorithmic bytes of SURVEY.md section 8d: mostly served by L1 / L2 / Infinity Cache when the scene fits them
            "algorithmic": {"bytes_per_launch": int(alg_bytes_per_launch), "GBs": round(alg_gbs, 1), "bytes_per_ray": round(alg_bytes_per_launch / max(1.0, rays_per_launch), 1),
                            "vs_counter_bytes": round(alg_bytes_per_launch / traffic, 2) if traffic else None,
                            "frac_of_l2_peak": round(alg_gbs / L2_PEAK_GBS, 4), "l2_peak_GBs": L2_PEAK_GBS,
                            "caches": {k: ck[k] for k in ("l1_hit_rate", "l2_hit_rate") if k in ck} if ck else None},
            "avg_launch_ms": round(avg_ms, 5), "launches": launches, "rays_per_launch": int(rays_per_launch),
            # where the durations come from: the timed repetitions' own frames, rendered again with event nodes around every kernel
            "timing": {
                "what": "event-record nodes around every kernel of the production graph; the image starts over, the warm-up frames are rendered again and every repetition of the "
                        "timed region is replayed on its own frame numbers; `avg_launch_ms`, `kernel_ms_per_frame` and every fraction of this block come from the repetition "
                        "whose wall time was the MEDIAN of the timed region (`repetition`, 0-based), i.e. from the frames `value` was measured on; a repetition of several passes: its last pass",
                "repetition": med_rep,
                "timed_region_rep_ms": [round(x * 1e3, 3) for x in rep_s],
                "instrumented_rep_ms": [round(t["wall_ms"], 3) for t in timed],
                "instrumented_replay_span_ms": [round(t["span_ms"], 3) for t in timed],
                "closest_hit_launch_ms_by_repetition": [[round(d0, 4) for kk, _, d0 in t["timeline"] if kk == "trace"] for t in timed],
                "critical_path_ms_per_frame": round(timed[med_rep]["span_ms"] / max(1, last_pass_frames), 4),
            },
            "rays_per_frame": closest["rays"] // frames, "nodes_per_ray": round(closest["nodes"] / max(1, closest["rays"]), 2),
            "tris_per_ray": round(closest["tris"] / max(1, closest["rays"]), 2),
            "instances_per_ray": round(closest["instances"] / max(1, closest["rays"]), 2),
            "simd": {"iters_per_ray_lane": round(64.0 * closest["waveIters"] / max(1, closest["rays"]), 2),
                     "active_frac": round(closest["lanesActive"] / max(1, 64 * closest["waveIters"]), 3),
                     "node_frac": round(closest["lanesNode"] / max(1, 64 * closest["waveIters"]), 3),
                     "prim_frac": round(closest["lanesPrim"] / max(1, 64 * closest["waveIters"]), 3),
                     "cycle_share": dict(zip(["refill", "pop_retire", "node_fetch", "node_decode", "instance", "triangle"],
                                             [round(c / max(1, sum(closest["cycles"])), 3) for c in closest["cycles"][:6]])),
                     "cycles_per_wave_iter": round(sum(closest["cycles"]) / max(1, closest["waveIters"]), 1)},
            "mrays_per_s": round(closest["rays"] / passes / (kt["trace"]["ms"] * 1e-3) / 1e6, 1) if kt["trace"]["ms"] > 0 else None,
            "shadow": {"rays_per_frame": shadow["rays"] // frames, "avg_launch_ms": round(kt["shadow"]["ms"] / max(1, kt["shadow"]["launches"]), 5),
                       "algorithmic_GBs": round((trace_algorithmic_bytes(shadow) / max(1, kt["shadow"]["launches"] * passes)) / max(1e-12, kt["shadow"]["ms"] / max(1, kt["shadow"]["launches"]) * 1e-3) / 1e9, 2)},
            "kernel_ms_per_frame": {k: round(v["ms"] / frames_timed[k], 4) for k, v in kt.items()},
            "items_per_frame": {"logic": int(sum(int(x) for x in q["traceSize"][: args.path_length]) // last_pass_frames) if kt["logic"]["launches"] else 0,
                                "shade": int(sum(int(sum(q[m][1: args.path_length + 1])) for m in ("diffuseSize", "plasticSize", "dielectricSize", "conductorSize")) // last_pass_frames)},
            "live_rays_by_bounce": [int(x) for x in q["traceSize"][: args.path_length + 1]],
        }

    # ---- the second kernel class: logic and the material (shade) kernels, per queue item
    if "roofline" in out:
        rf = out["roofline"]
        classes = {}
        # SURVEY.md section 8d, algorithmic bytes per item.  Logic: 60 in + 48 out, + 164 gathered (instance 160 + material 4)
        # for a hit.  Shade: 36 request + 12 throughput + 348 gathered (instance 160, BVH descriptor 32, triangle 96, material
        # 60) + 360 for the light sample, out 28 + 44 + 28 + 24.
        alg = {"logic": 60 + 48 + 164, "shade": 36 + 12 + 348 + 360 + 28 + 44 + 28 + 24}
        for klass in ("logic", "shade"):
            items = rf["items_per_frame"][klass]
            ms = rf["kernel_ms_per_frame"].get(klass)
            if not items or not ms:
                continue
            sec = ms * 1e-3
            e = {"items_per_frame": items, "ms_per_frame": ms, "Mitems_per_s": round(items / sec / 1e6, 1), "algorithmic_bytes_per_item": alg[klass],
                 "algorithmic_GBs": round(items * alg[klass] / sec / 1e9, 1)}
            ck2, why = counters_for(args.config, klass, "wavefront")
            if ck2:
                bytes_item = ck2.get("hbm_read_bytes_per_ray", 0.0) + ck2.get("hbm_write_bytes_per_ray", 0.0)
                gbs = items * bytes_item / sec / 1e9
                e["hbm"] = {"bytes_per_item": round(bytes_item, 1), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                            "frac_of_achievable_6290": round(gbs / 6290.0, 4)}
                if "valu_insts_per_ray" in ck2:
                    ginst = items * ck2["valu_insts_per_ray"] / sec / 1e9
                    peak_ginst = SIMDS * ck2.get("clock_GHz", 2.3) / VALU_ISSUE_CYCLES
                    e["valu-issue"] = {"valu_insts_per_item": ck2["valu_insts_per_ray"], "achieved": round(ginst, 1), "peak": round(peak_ginst, 1), "unit": "G wave-instructions/s",
                                       "frac": round(ginst / peak_ginst, 4)}
                for k in ("l1_hit_rate", "l2_hit_rate", "valu_active_lanes_per_inst", "vmem_rd_insts_per_ray"):
                    if k in ck2:
                        e[k] = ck2[k]
                e["bound"] = max((k for k in ("hbm", "valu-issue") if k in e), key=lambda k: e[k]["frac"])
            else:
                e["counters"] = why
            classes[klass] = e
        rf["classes"] = classes

    if rank == 0 and not dist_mode and not args.no_cpu_baseline:
        # a 1-GPU box's CPU share is 16 hardware threads
        stamp("cpu baseline (oracle on the host cores)")
        threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        cb = cpu_baseline(sc, W, H, threads, 2 if big else 100 if cornell else 12, single_core=not big)  # about 10-15 s of CPU work on the box's 16 threads
        cb["value"] = round(cb["value"], 4)
        if cornell:
            cb["bvh2_primary_rays"] = cpu_bvh2_baseline(sc, W, H, threads, 100)
        out["cpu_baseline"] = cb

    if rank == 0 and args.png:
        from nexus_amd import imageio

        if dist_mode:
            img = headline_full.cpu().numpy().view(np.uint32)
        elif args.pixel_order == "tiles":
            img = np.zeros(W * H, np.uint32)
            img[multigpu.tiled_order(np.arange(W * H, dtype=np.uint32), W)] = headline_rgba8
        else:
            img = headline_rgba8
        imageio.write_png(args.png, img, W, H)

    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    stamp("done")
    import faulthandler
    faulthandler.cancel_dump_traceback_later()
    if dist_mode:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
