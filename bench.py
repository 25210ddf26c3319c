#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s of the Cornell box, 1920x1080 at 4096 spp (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE). A "step" is one full render of the
metric workload: the framebuffer is split into N row tiles, rank g path-traces rows
[g*H/N,(g+1)*H/N) for all spp through the C ABI (librtw_hip.so, hand-written HIP), then ONE RCCL
gather brings the float4 tiles to rank 0. Total work is fixed as N grows ("strong" scaling).
Rank 0 prints ONE JSON line. torch is plumbing only: device memory, stream, torch.distributed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, DEPTH, SEED = 1920, 1080, 4096, 50, 0x6314759
CONFIGS = {"headline": (1920, 1080, 4096), "c5": (7680, 4320, 4096)}  # BASELINE.json: the metric workload, and config 5
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # wave64 VALU instructions per ns, chip-wide: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles (same guide)


def cpu_baseline(blob, abi):
    """CPU oracle (oracle/rtw_oracle.c, kind "port") timed on this host's cores on a BOUNDED sample of
    the same workload: the full 1920x1080 frame at depth 50, at the few spp that take about 10 s.
    Baseline only: it says nothing about kernel quality (the roofline fraction does)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle  # the checker; used here only as the timed CPU baseline
    cores = max(1, min(os.cpu_count() or 1, 64))

    def timed(p, threads):
        t = time.perf_counter()
        _, st = oracle.render(blob, p, threads=threads)
        return st, time.perf_counter() - t

    # calibrate on one spp, then size each timed run to ~10 s of wall time
    st, dt = timed(abi.make_params(WIDTH, HEIGHT, 1, DEPTH, seed=SEED), cores)
    spp_n = int(max(1, min(256, round(10.0 / max(dt, 1e-3)))))
    sn, dtn = timed(abi.make_params(WIDTH, HEIGHT, spp_n, DEPTH, seed=SEED), cores)
    band = (472, 607)  # 135 rows through the middle of the frame
    st1, dt1 = timed(abi.make_params(WIDTH, HEIGHT, 1, DEPTH, seed=SEED, row0=band[0], row1=band[1]), 1)
    spp_1 = int(max(1, min(256, round(10.0 / max(dt1, 1e-3)))))
    s1, dt1 = timed(abi.make_params(WIDTH, HEIGHT, spp_1, DEPTH, seed=SEED, row0=band[0], row1=band[1]), 1)
    return {
        "value": round(sn.samples / dtn / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"Cornell box {WIDTH}x{HEIGHT}, {spp_n} spp, depth {DEPTH}, {cores} threads, {dtn:.1f} s; "
                  f"single thread: rows {band[0]}-{band[1] - 1} at {spp_1} spp, {dt1:.1f} s",
        "single_thread_value": round(s1.samples / dt1 / 1e6, 4),
        "segments_per_sample": round(sn.segments / sn.samples, 4),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS), help="headline: Cornell box 1920x1080 4096 spp (the metric); "
                    "c5: BASELINE config 5, Cornell box 7680x4320 4096 spp (row shards over the GPUs)")
    ap.add_argument("--spp", type=int, default=None, help="override samples per pixel (a reduced-spp line is not the headline metric)")
    ap.add_argument("--rng", type=int, default=0, help="0 Philox4x32-10 (default), 1 the reference's TEA+LCG")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1: nccl (= RCCL over xGMI) or gloo "
                                                    "(rehearsal on one GPU: ranks share device 0, tiles are gathered through host memory)")
    ap.add_argument("--check", action="store_true", help="rank 0 re-renders the full frame alone and asserts the gathered image is identical")
    args = ap.parse_args()

    global WIDTH, HEIGHT
    WIDTH, HEIGHT, spp_cfg = CONFIGS[args.config]
    if args.spp is None:
        args.spp = spp_cfg
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs dmabuf IPC on this pool
    import torch
    import torch.distributed as dist
    from raytracing_weekend_amd import abi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start N > 1 ranks with `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...` (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
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

    blob = abi.build_scene(0, WIDTH, HEIGHT)
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
            # the one collective: row tiles -> rank 0 (RCCL over xGMI; through host memory in the gloo rehearsal)
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
    kt = torch.tensor([sum(s.bounce_seconds for s in stats), float(sum(s.bounce_launches for s in stats)),
                       sum(s.seconds for s in stats)], dtype=torch.float64, device=dev)
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
            raise SystemExit("gathered row tiles differ from the single-tile render")
    if rank == 0:
        # Dominant kernel of rank 0 = the path-tracing kernel with the largest summed device time (k_path on the metric
        # workload). Its algorithmic bytes are 128 B per radiance segment it processed (64 B of path state read
        # and written once per segment; for k_trace, per ray pair traced); its time is measured live with HIP events
        # recorded on the launch stream around every launch inside rtw_render_device.
        names = abi.Stats.KERNELS
        NK = len(names)
        k_s = [sum(s.kernel_seconds[i] for s in stats) for i in range(NK)]
        k_n = [sum(s.kernel_launches[i] for s in stats) for i in range(NK)]
        k_seg = [sum(s.kernel_segments[i] for s in stats) for i in range(NK)]
        dom = max(range(NK), key=lambda i: k_s[i])
        seg0 = float(sum(s.segments for s in stats))
        b_s, b_n, r_s = (float(x) for x in kt.tolist())
        dom_units = float(k_seg[dom]) if names[dom] != "k_trace" else float(k_seg[dom]) / 2.0
        achieved = 128.0 * dom_units / k_s[dom] / 1e9 if k_s[dom] > 0 else 0.0
        loop_achieved = 128.0 * seg0 / b_s / 1e9 if b_s > 0 else 0.0
        # Measured HBM traffic and instruction counts PER SEGMENT of that kernel come from the committed rocprofv3 --pmc
        # summary (profiles/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE and SQ_INSTS_VALU of a profiled run of the same
        # kernel on the same scene); they are scaled here by the units this run's launches processed.
        traffic, pmc_rec, valu = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    pmc_rec = json.load(f).get(names[dom])
                if pmc_rec and k_n[dom]:
                    traffic = int(pmc_rec["hbm_bytes_per_segment"] * dom_units / k_n[dom])
                    if pmc_rec.get("valu_insts_per_segment") and k_s[dom] > 0:
                        rate = pmc_rec["valu_insts_per_segment"] * dom_units / k_s[dom] / 1e9
                        valu = {"wave_insts_per_64_segments": round(64.0 * pmc_rec["valu_insts_per_segment"], 1),
                                "achieved_Ginst_per_s": round(rate, 1), "peak_Ginst_per_s": VALU_PEAK_GINST,
                                "frac": round(rate / VALU_PEAK_GINST, 4), "lane_utilisation": pmc_rec.get("lane_utilisation")}
            except Exception:
                traffic = None
        in_regs = names[dom] == "k_path"
        cfg_name = "Cornell box 1920x1080 4096spp" if args.config == "headline" else "Cornell box 7680x4320 4096spp (BASELINE config 5)"
        line = {
            "metric": f"Msamples/s, {cfg_name}",
            "value": round(samples / dt_max / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Cornell box (reference scene 0) {WIDTH}x{HEIGHT}, {args.spp} spp, max depth {DEPTH}, "
                                   f"NEE mixture PDF (cosine + light rect), RR from depth 2, "
                                   f"{'Philox4x32-10' if args.rng == 0 else 'TEA+LCG'} seed 0x{SEED:x}",
                       "partition": f"{world} interleaved row shards (rank g: rows g, g+{world}, ...; {max_rows} rows each), one RCCL gather per step" if world > 1 else "single tile",
                       "segments_per_sample": round(segments / samples, 4),
                       "shadow_rays_per_sample": round(shadow / samples, 4),
                       "pipeline": "k_path: paths in registers, lanes own (pixel, 4 x 16-sample block) units and regenerate; one bulk launch + a concurrent fine-grained end-game launch per pass" if in_regs
                                   else "wavefront: k_first / k_trace / k_shade / k_bounce over SoA path state in HBM"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json (static: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE per segment of a "
                                           "profiled run, x this run's segments per launch)" if traffic is not None else None,
                         "kernel": names[dom], "launches": int(k_n[dom]),
                         "avg_launch_us": round(k_s[dom] / k_n[dom] * 1e6, 3) if k_n[dom] else None,
                         "algorithmic_bytes_per_launch": round(128.0 * dom_units / k_n[dom], 1) if k_n[dom] else None,
                         "note": ("achieved = 128 B x segments / kernel time (SURVEY 8d; the pass's two overlapping k_path launches are timed as one, from before the first to after both). k_path never writes path state: the state the "
                                  "algorithmic figure counts stays in registers, only 16 B per pixel and 64 samples reach HBM (traffic), so "
                                  "the kernel is bound by VALU issue, not by HBM: see roofline.valu") if in_regs else
                                 "two batches are in flight on two streams (lanes), so a kernel's own launch duration includes sharing "
                                 "the GPU with the other lane's kernels; whole_loop is the unshared figure",
                         "valu": valu,
                         "whole_loop": {"achieved": round(loop_achieved, 2), "frac": round(loop_achieved / HBM_PEAK_GBS, 5),
                                        "seconds": round(b_s, 4), "launches": int(b_n),
                                        "note": "128 B x all segments / device time of the render calls (every kernel, resolve included)"},
                         "per_kernel": {names[i]: {"seconds": round(k_s[i], 4), "launches": int(k_n[i]), "units": int(k_seg[i])}
                                        for i in range(NK) if k_n[i]},
                         "render_device_seconds_rank0": round(r_s, 4)},
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
            line["cpu_baseline"] = cpu_baseline(blob, abi)
        print(json.dumps(line), flush=True)
    r.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
