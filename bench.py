#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s of the Cornell box, 1920x1080 at 4096 spp (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config headline|c2|c3|c4|c5]

One process per GPU. Started as `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment, this
process starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a fresh CHILD before it imports
torch or touches the GPU, relays the child's JSON line and exits with its code (no exec of a process that has
initialised the GPU). Under torchrun (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank.

A "step" is one full render of the configuration: the framebuffer is split into N interleaved row shards, rank g
path-traces rows g, g+N, ... for all spp through the C ABI (librtw_hip.so, hand-written HIP), then ONE gather (RCCL
over xGMI) brings the float4 shards to rank 0. Total work is fixed as N grows ("strong" scaling).
Rank 0 prints ONE JSON line. torch is plumbing only: device memory, stream, torch.distributed.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEPTH, SEED = 50, 0x6314759
# BASELINE.json: the metric workload and configs[1..4] (configs[0] is the CPU-runnable golden fixture: a parity test, not a bench line)
CONFIGS = {
    "headline": dict(scene=0, w=1920, h=1080, spp=4096, name="Cornell box 1920x1080 4096spp",
                     what="Cornell box (reference scene 0)", light="NEE mixture PDF (cosine + light rect)"),
    "c2": dict(scene=0, w=800, h=800, spp=1024, name="Cornell box 800x800 1024spp (BASELINE config 2)",
               what="Cornell box (reference scene 0)", light="NEE mixture PDF (cosine + light rect)"),
    "c3": dict(scene=1, w=1920, h=1080, spp=512, name="random spheres 1920x1080 512spp (BASELINE config 3)",
               what="Book-1 random spheres (reference scene 1, 528 primitives, 333 moving; 4-wide tree)", light="sky light, no listed lights"),
    "c4": dict(scene=3, w=1920, h=1080, spp=2048, name="Cornell box + fog 1920x1080 2048spp (BASELINE config 4)",
               what="Cornell box with two constant media (reference scene 3)", light="sky light, no listed lights (SURVEY Q11)"),
    "c5": dict(scene=0, w=7680, h=4320, spp=4096, name="Cornell box 7680x4320 4096spp (BASELINE config 5)",
               what="Cornell box (reference scene 0)", light="NEE mixture PDF (cosine + light rect)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # wave64 VALU instructions per ns, chip-wide: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles (same guide)
PIPELINES = {
    "k_path": "k_path: paths in registers, lanes own (pixel, run of 16-sample blocks) units and regenerate; one bulk launch + a concurrent fine-grained end-game launch per pass",
    "wavefront": "wavefront: k_first / k_trace_bvh / k_shade / k_bounce over SoA path state in HBM, two batches in flight on two streams",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS),
                    help="headline: Cornell box 1920x1080 4096 spp (the metric); c2 / c3 / c4 / c5: BASELINE.json configs 2-5")
    ap.add_argument("--spp", type=int, default=None, help="override samples per pixel (a reduced-spp line is not the configuration's metric)")
    ap.add_argument("--rng", type=int, default=0, help="0 Philox4x32-10 (default), 1 the reference's TEA+LCG")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1: nccl (= RCCL over xGMI) or gloo "
                                                    "(rehearsal on one GPU: ranks share device 0, shards are gathered through host memory)")
    ap.add_argument("--check", action="store_true", help="rank 0 re-renders the full frame alone and asserts the gathered image is identical")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a fresh child (this process has not imported
    torch and never touches the GPU), pass the child's stderr through, relay its JSON line, return its exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    for line in proc.stdout:  # the JSON line goes to stdout, whatever else the ranks' libraries print there (gloo does) to stderr
        out = sys.stdout if line.startswith('{"metric"') else sys.stderr
        out.write(line)
        out.flush()
    return proc.wait()


def cpu_baseline(blob, abi, cfg):
    """CPU oracle (oracle/rtw_oracle.c, kind "port") timed on this host's cores on a BOUNDED sample of the same
    workload: the configuration's full frame at depth 50, at the few spp that take about 10 s.
    Baseline only: it says nothing about kernel quality (the roofline fraction does)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle  # the checker; used here only as the timed CPU baseline
    W, H = cfg["w"], cfg["h"]
    cores = max(1, min(os.cpu_count() or 1, 64))

    def timed(p, threads):
        t = time.perf_counter()
        _, st = oracle.render(blob, p, threads=threads)
        return st, time.perf_counter() - t

    # calibrate on one spp, then size each timed run to ~10 s of wall time
    st, dt = timed(abi.make_params(W, H, 1, DEPTH, seed=SEED), cores)
    spp_n = int(max(1, min(256, round(10.0 / max(dt, 1e-3)))))
    sn, dtn = timed(abi.make_params(W, H, spp_n, DEPTH, seed=SEED), cores)
    band = (H // 2 - H // 16, H // 2 + H // 16)  # an eighth of the rows through the middle of the frame
    st1, dt1 = timed(abi.make_params(W, H, 1, DEPTH, seed=SEED, row0=band[0], row1=band[1]), 1)
    spp_1 = int(max(1, min(256, round(10.0 / max(dt1, 1e-3)))))
    s1, dt1 = timed(abi.make_params(W, H, spp_1, DEPTH, seed=SEED, row0=band[0], row1=band[1]), 1)
    return {
        "value": round(sn.samples / dtn / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"{cfg['what']} {W}x{H}, {spp_n} spp, depth {DEPTH}, {cores} threads, {dtn:.1f} s; "
                  f"single thread: rows {band[0]}-{band[1] - 1} at {spp_1} spp, {dt1:.1f} s",
        "single_thread_value": round(s1.samples / dt1 / 1e6, 4),
        "segments_per_sample": round(sn.segments / sn.samples, 4),
    }


def pmc_record(config, kernel):
    """Per-unit HBM bytes and instruction counts of `kernel` on this configuration's scene, from the committed rocprofv3 --pmc
    summary (profiles/pmc_traffic.json: {config: {kernel: {...}}}; FETCH_SIZE x 2 + WRITE_SIZE, separate passes)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return None
    key = {"c2": "headline", "c5": "headline"}.get(config, config)  # the Cornell configurations run the same kernel on the same scene
    return (table.get(key) or {}).get(kernel)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    cfg = dict(CONFIGS[args.config])
    WIDTH, HEIGHT = cfg["w"], cfg["h"]
    if args.spp is None:
        args.spp = cfg["spp"]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs dmabuf IPC on this pool
    import torch
    import torch.distributed as dist
    from raytracing_weekend_amd import abi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if world > n_dev and args.backend == "nccl":
        raise SystemExit(f"--gpus {world} with backend nccl needs {world} GPUs, this node shows {n_dev} (RCCL refuses two ranks on one "
                         f"device; `--backend gloo` rehearses the N-rank path on fewer)")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RTW_BENCH_FORCE_DIST=1: open the process group and run the gather even at world size 1 (a rehearsal of the RCCL path on a
    # one-GPU box: two ranks cannot share a device under RCCL, one rank can still initialise it and gather to itself)
    use_dist = world > 1 or os.environ.get("RTW_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    blob = abi.build_scene(cfg["scene"], WIDTH, HEIGHT)
    from raytracing_weekend_amd.dist import interleaved_shard
    # interleaved rows: rank g renders rows g, g+N, g+2N ... (balanced; contiguous tiles of a Cornell box are not)
    row0, row1, row_stride, my_rows = interleaved_shard(HEIGHT, world, rank)
    n_rows = [interleaved_shard(HEIGHT, world, g)[3] for g in range(world)]
    max_rows = max(n_rows)
    params = abi.make_params(WIDTH, HEIGHT, args.spp, DEPTH, seed=SEED, row0=row0, row1=row1, rng_kind=args.rng,
                             row_stride=row_stride)

    r = abi.Renderer(dev_index)
    r.upload_scene(blob)
    tile = torch.zeros((max_rows, WIDTH, 4), dtype=torch.float32, device=dev)
    host_gather = use_dist and args.backend != "nccl"
    gdev = torch.device("cpu") if host_gather else dev
    gathered = [torch.empty((max_rows, WIDTH, 4), dtype=torch.float32, device=gdev) for _ in range(world)] if (use_dist and rank == 0) else None
    # a stream of our own: handle 0 (torch's default stream) would select the library's private stream (include/rtw.h)
    tstream = torch.cuda.Stream(dev)
    tstream.wait_stream(torch.cuda.current_stream(dev))
    stream = tstream.cuda_stream

    def step():
        st = r.render_device(params, tile.data_ptr(), stream)
        if use_dist:
            # the one collective: row shards -> rank 0 (RCCL over xGMI; through host memory in the gloo rehearsal)
            dist.gather(tile.cpu() if host_gather else tile, gathered, dst=0)
        return st

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    fence()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    agg = torch.tensor([float(sum(s.segments for s in stats)), float(sum(s.samples for s in stats)),
                        float(sum(s.shadow_rays for s in stats))], dtype=torch.float64, device=dev)
    if host_gather:
        t, agg = t.cpu(), agg.cpu()
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    dt_max = float(t.item())
    segments, samples, shadow = (float(x) for x in agg.tolist())

    if args.check and rank == 0:
        full = torch.empty((HEIGHT, WIDTH, 4), dtype=torch.float32)
        for g in range(world):
            full[g::world] = (gathered[g] if use_dist else tile)[: n_rows[g]].cpu()
        alone = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev)
        r.render_device(abi.make_params(WIDTH, HEIGHT, args.spp, DEPTH, seed=SEED, rng_kind=args.rng), alone.data_ptr(), stream)
        torch.cuda.synchronize(dev)
        if not torch.equal(full, alone.cpu()):
            raise SystemExit("gathered row shards differ from the single-tile render")
    if rank == 0:
        # Dominant kernel of rank 0 = the path-tracing kernel with the largest summed device time (k_path on the Cornell
        # configurations, k_trace_bvh on the tree scene). Algorithmic bytes: 128 B per unit it processed (64 B of
        # path state read and written once per radiance segment; a trace kernel's unit is a path slot = the segment's radiance
        # ray and its queued probe). The 32 B per finished sample of SURVEY 8d belong to the resolve kernels and are NOT in the
        # dominant kernel's figure; whole_loop states both forms. Time: HIP events recorded on the launch stream around every
        # launch inside rtw_render_device.
        names = abi.Stats.KERNELS
        NK = len(names)
        k_s = [sum(s.kernel_seconds[i] for s in stats) for i in range(NK)]
        k_n = [sum(s.kernel_launches[i] for s in stats) for i in range(NK)]
        k_u = [sum(s.kernel_segments[i] for s in stats) for i in range(NK)]
        dom = max(range(NK), key=lambda i: k_s[i])
        seg0 = float(sum(s.segments for s in stats))
        smp0 = float(sum(s.samples for s in stats))
        r_s = sum(s.seconds for s in stats)
        n_launch = sum(s.bounce_launches for s in stats)
        dom_units = float(k_u[dom])
        achieved = 128.0 * dom_units / k_s[dom] / 1e9 if k_s[dom] > 0 else 0.0
        loop128 = 128.0 * seg0 / r_s / 1e9 if r_s > 0 else 0.0
        loop_full = (128.0 * seg0 + 32.0 * smp0) / r_s / 1e9 if r_s > 0 else 0.0
        in_regs = names[dom] == "k_path"

        def per_kernel_pmc(i):
            """(traffic bytes per launch, valu block) of kernel i from the committed counters, scaled by this run's units"""
            rec = pmc_record(args.config, names[i])
            if not rec or not k_n[i]:
                return None, None, rec
            traffic = int(rec["hbm_bytes_per_unit"] * k_u[i] / k_n[i]) if rec.get("hbm_bytes_per_unit") is not None else None
            valu = None
            if rec.get("valu_insts_per_unit") and k_s[i] > 0:
                rate = rec["valu_insts_per_unit"] * k_u[i] / k_s[i] / 1e9
                valu = {"wave_insts_per_64_units": round(64.0 * rec["valu_insts_per_unit"], 1),
                        "achieved_Ginst_per_s": round(rate, 1), "peak_Ginst_per_s": VALU_PEAK_GINST,
                        "frac": round(rate / VALU_PEAK_GINST, 4), "lane_utilisation": rec.get("lane_utilisation")}
            return traffic, valu, rec
        traffic, valu, pmc_rec = per_kernel_pmc(dom)
        # what binds the dominant kernel: k_path keeps its path state in registers, so HBM is idle and the VALU issue rate is
        # the limit; the wavefront kernels stream state through HBM
        bound = "valu" if in_regs else "hbm"
        roof = {"bound": bound, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": (pmc_rec or {}).get("source") if traffic is not None else None,
                "kernel": names[dom] + ("_bvh" if names[dom] == "k_trace" and cfg["scene"] in (1, 2, 4) else ""),
                "launches": int(k_n[dom]),
                "avg_launch_us": round(k_s[dom] / k_n[dom] * 1e6, 3) if k_n[dom] else None,
                "algorithmic_bytes_per_launch": round(128.0 * dom_units / k_n[dom], 1) if k_n[dom] else None,
                "algorithmic_equiv": "achieved / frac = 128 B x the kernel's units / its summed launch time (SURVEY 8d; the 32 B per finished "
                                     "sample are the resolve kernels' and are not included; whole_loop gives both forms)",
                "valu": valu}
        if in_regs:
            roof["bound_frac"] = valu["frac"] if valu else None
            roof["note"] = ("k_path never writes path state: the 64 B the algorithmic figure counts stay in registers and only 16 B per pixel "
                            "and 128-sample summation unit (per 16-sample block in the end-game region and in short renders) reach HBM "
                            "(traffic), so HBM is idle and the kernel is bound by VALU issue: bound_frac = "
                            "roofline.valu.frac is the physical fraction, frac the north star's HBM-equivalent figure of merit. The pass's two "
                            "overlapping k_path launches are timed as one, from before the first to after both")
        else:
            roof["note"] = ("two batches are in flight on two streams (lanes), so a kernel's own launch duration includes sharing the GPU "
                            "with the other lane's kernels; whole_loop is the unshared figure")
        roof["whole_loop"] = {"achieved": round(loop128, 2), "frac": round(loop128 / HBM_PEAK_GBS, 5),
                              "achieved_with_32B_per_sample": round(loop_full, 2), "frac_with_32B_per_sample": round(loop_full / HBM_PEAK_GBS, 5),
                              "seconds": round(r_s, 4), "launches": int(n_launch),
                              "note": "128 B x all segments (second form: + 32 B x samples) / device time of the render calls (every kernel, resolve included)"}
        per = {}
        for i in range(NK):
            if not k_n[i]:
                continue
            tr_i, valu_i, _ = per_kernel_pmc(i)
            a_i = 128.0 * k_u[i] / k_s[i] / 1e9 if k_s[i] > 0 else 0.0
            per[names[i]] = {"seconds": round(k_s[i], 4), "launches": int(k_n[i]), "units": int(k_u[i]),
                             "achieved_GBps": round(a_i, 1), "frac": round(a_i / HBM_PEAK_GBS, 4),
                             "traffic_bytes_per_unit": round(tr_i * k_n[i] / k_u[i], 2) if (tr_i is not None and k_u[i]) else None,
                             "valu_frac": valu_i["frac"] if valu_i else None}
        roof["per_kernel"] = per
        roof["render_device_seconds_rank0"] = round(r_s, 4)
        line = {
            "metric": f"Msamples/s, {cfg['name']}",
            "value": round(samples / dt_max / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{cfg['what']} {WIDTH}x{HEIGHT}, {args.spp} spp, max depth {DEPTH}, {cfg['light']}, RR from depth 2, "
                                   f"{'Philox4x32-10' if args.rng == 0 else 'TEA+LCG'} seed 0x{SEED:x}",
                       "name": args.config,
                       "partition": f"{world} interleaved row shards (rank g: rows g, g+{world}, ...; {max_rows} rows each), one "
                                    f"{'RCCL' if args.backend == 'nccl' else args.backend} gather per step" if world > 1 else "single tile",
                       "segments_per_sample": round(segments / samples, 4),
                       "shadow_rays_per_sample": round(shadow / samples, 4),
                       "Msegments_per_s": round(segments / dt_max / 1e6, 1),
                       "pipeline": PIPELINES["k_path" if in_regs else "wavefront"]},
            "roofline": roof,
        }
        if world == 1:
            # what a plain device-to-device copy of 2 GiB reaches on this very GPU, measured after the timed region:
            # the practical ceiling for a read+write stream, to read the fractions of the 8 TB/s spec peak against
            try:
                src = torch.empty(1 << 29, dtype=torch.float32, device=dev)
                dst = torch.empty_like(src)
                dst.copy_(src); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    dst.copy_(src)
                e1.record(); torch.cuda.synchronize()
                line["roofline"]["device_copy_GBps"] = round(5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
                del src, dst
            except Exception as ex:  # measurement aid only
                line["roofline"]["device_copy_GBps"] = None
                print(f"[bench] device copy probe failed: {ex}", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(blob, abi, cfg)
        print(json.dumps(line), flush=True)
    r.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
