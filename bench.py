#!/usr/bin/env python3
"""bench.py -- DSM tiles/sec through one full GAN train step (G + D + losses + Adam).

Default workload = BASELINE.json configs[1]: 256x256 1-channel synthetic DSM tiles + random disc masks, batch 16
PER GPU, fp32, partial-conv U-Net generator + PatchGAN discriminator + L1/perceptual/TV/boundary losses.
A "step" is one pass of mvp_gan.src.train.train_step over one batch that is already resident in HBM.
Other BASELINE configs: --size 512 --batch 8 --precision bf16 (configs[2]); --size 1024 --batch 4 --checkpoint
(configs[4] per GPU); `metric` and `config.workload` are derived from the arguments.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = tiles processed by all ranks / max-over-ranks wall time of exactly K
steps (barrier + synchronize on both sides).  `roofline` is measured live: an extra instrumented pass after the
timed region brackets every launch of the dominant kernel (the fp32-MFMA Winograd conv) with hipEvents on
its launch stream (tg_prof_*), achieved = sum of algorithmic FLOPs / sum of kernel time.  `cpu_baseline` times
the CPU oracle (oracle/terragan_oracle.py, "port") on the host cores, rank 0 at N=1 only.

Roofline fields (every one can be recomputed from profiles/*_conv_launches.csv + *_kernel_stats.csv):
  achieved          EXECUTED MFMA TFLOP/s of the dominant kernel = algorithmic FLOPs / 2.25 / kernel time (Winograd
                    F(2x2,3x3) issues 16 multiplies where the direct convolution has 36; the VGG trunk's F(4x4,3x3) kernel,
                    listed under other_kernels, issues 36 per 16 outputs = 1/4)
  frac              achieved / 157.3 TF (fp32-MFMA dense peak) -- the share of the matrix pipe that is busy, <= 1
  effective_tflops  algorithmic (direct-convolution) FLOPs / kernel time -- what the layer would need on a direct kernel
  layers            per-layer lines for the full-resolution PConv layers the 40 %-of-HBM target is about (enc1, dec1,
                    dec2, final) and the discriminator's 4x4 stride-2 layers (d2, d5, d8: Winograd F(2x2,2x2), 9 multiplies
                    where the direct convolution has 16): alg_GBps / frac_hbm against 8 TB/s next to tflops / frac_mfma
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "terra-gan_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: bf16 MFMA dense (not the 2:1-sparsity headline)
PEAK_HBM_GBS = 8000.0
TILE = 256
BATCH = 16


def prof_summary(lib, kind):
    ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    lib.tg_prof_summary(kind, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
    return ms.value, n.value, fl.value, by.value


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(size=TILE, batch=BATCH, seconds_budget=45.0):
    """CPU oracle train step on the host cores, bounded sample of the SAME workload: the bench's own batch size and tile
    size (B=16 at 256x256, 2 warm-up + 5 timed steps; B=1 = configs[0] is timed beside it).  The
    step count only drops below 5 when a step is so slow that 5 of them would exceed ~seconds_budget (larger tiles)."""
    from oracle import terragan_oracle as Orc
    # the GPU box gives a 1-GPU job a 16-core CPU share although os.cpu_count() reports the whole host
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))

    def timed(b, budget, max_steps):
        st = Orc.TrainState(0)
        real, mask = Orc.synth_batch(b, size, 1000)
        for _ in range(2):
            Orc.train_step(st, real, mask)              # warm-up (oneDNN primitive creation, allocator)
        t0, n = time.perf_counter(), 0
        while True:
            Orc.train_step(st, real, mask)
            n += 1
            el = time.perf_counter() - t0
            if n >= max_steps or el * (n + 1) / n > budget:
                break
        return n, el

    n, el = timed(batch, seconds_budget * 0.8, 5)
    n1, el1 = timed(1, seconds_budget * 0.2, 5)
    return {"value": round(batch * n / el, 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": _cpu_model(), "b1_value": round(n1 / el1, 3),
            "sample": f"{n} train steps of the CPU oracle at batch {batch}, {size}x{size} fp32, after 2 warm-up steps "
                      f"({el:.1f} s); b1_value = {n1} steps at batch 1 (BASELINE configs[0], {el1:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (default: BASELINE config 2)")
    ap.add_argument("--size", type=int, default=TILE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"],
                    help="conv inner-product arithmetic: f32 (BASELINE config 2, default) or bf16 operands + fp32 accumulate (config 3)")
    ap.add_argument("--checkpoint", action="store_true", help="activation checkpointing in the generator (config 5)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the train step in a hipGraph and replay it (tg_hip.graph; single GPU; pays off at small batches)")
    ap.add_argument("--prof-dump", default=None, help="write the per-launch table of the instrumented pass to this CSV")
    args = ap.parse_args()

    # before the first HIP call (torch.cuda.is_available() initialises the runtime): the host driver only supports dmabuf IPC
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("TERRAGAN_ALLOW_STANDIN_VGG", "1")      # no network: stand-in VGG16 weights (same FLOPs), see `data`
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # rehearsal switches (never used by the driver): all ranks on device 0 + gloo transport, to exercise the
    # multi-rank control flow on a single-GPU box
    backend = os.environ.get("TG_DIST_BACKEND", "nccl")
    if os.environ.get("TG_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dp = os.environ.get("TG_FORCE_DP") == "1"        # exercise the RCCL path with a single rank (rehearsal)
    if world > 1 or force_dp:
        if force_dp and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
            else:
                dist.init_process_group(backend)

    from mvp_gan.src.models import Discriminator, PConvUNet
    from mvp_gan.src.train import train_step
    from mvp_gan.src.utils.losses import InpaintingLoss
    from tg_hip.synth import synth_batch                   # seeded synthetic inputs (SURVEY §8d recipe)
    from tg_hip import lib as L
    from tg_hip.dist import GradSync
    lib = L.load()
    from tg_hip import ops as _O
    _O.set_precision(args.precision)

    torch.manual_seed(0)                                    # identical weights on every rank
    G, D = PConvUNet(), Discriminator()
    crit = InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
    G, D, crit = G.to(dev), D.to(dev), crit.to(dev)
    oG = torch.optim.Adam(G.parameters(), lr=2e-4)
    oD = torch.optim.Adam(D.parameters(), lr=2e-4)
    G.train(), D.train()
    G.activation_checkpointing = args.checkpoint
    sync = GradSync(world) if (world > 1 or force_dp) else None

    nb = 4                                                  # distinct resident batches, cycled
    batches = []
    for i in range(nb):
        real, mask = synth_batch(args.batch, args.size, 1000 + i * world + rank)
        batches.append((real.to(dev), mask.to(dev)))

    gstep = None
    if args.graph:
        assert sync is None, "--graph is single-GPU"
        from tg_hip.graph import GraphedTrainStep
        gstep = GraphedTrainStep(G, D, crit, oG, oD, warmup=2)

    def run(k):
        for i in range(k):
            real, mask = batches[i % nb]
            if gstep is not None:
                gstep(real, mask)
            else:
                train_step(G, D, crit, oG, oD, real, mask, grad_sync=sync)

    run(max(args.warmup, 3) if gstep is not None else args.warmup)      # the capture (3rd call) stays outside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    if gstep is not None:
        gstep.flush()
        gstep = None            # the instrumented pass below runs eagerly (per-launch events cannot be recorded inside a graph)
    roofline = None
    if not args.no_roofline:
        # instrumented pass (outside the timed region): hipEvents around every MFMA conv launch, on its launch stream.
        # Every rank runs the probe steps (they contain the gradient all-reduce); only rank 0 records.
        import csv
        import tempfile
        nprobe = 2
        if rank == 0:
            lib.tg_prof_enable(1)
            _O.PROF_TAGS = True
        run(nprobe)
        torch.cuda.synchronize()
        lib.tg_prof_enable(0)
        _O.PROF_TAGS = False
    if rank == 0 and not args.no_roofline:
        dump = args.prof_dump or os.path.join(tempfile.gettempdir(), f"tg_prof_{os.getpid()}.csv")
        lib.tg_prof_dump(dump.encode())
        rows = list(csv.DictReader(open(dump)))
        for kind in (0, 1, 2, 3):
            prof_summary(lib, kind)                       # consume the records

        WINO = ("4064", "4164", "4016")                   # Winograd F(2x2,3x3) kernels execute 16/36 of the algorithmic multiplies
        WINO22 = ("4022", "4122")                         # F(2x2,2x2) (the 4x4 stride-2 convs of D): 9/16
        WINO44 = ("4044",)                                # F(4x4,3x3) (the frozen VGG trunk): 36 multiplies per 16 outputs = 1/4

        def executed(r):
            return float(r["gflop"]) / (2.25 if r["cfg"] in WINO else 16.0 / 9.0 if r["cfg"] in WINO22 else 4.0 if r["cfg"] in WINO44 else 1.0)

        def agg(pred):
            sel = [r for r in rows if pred(r)]
            ms = sum(float(r["ms"]) for r in sel)
            fl = sum(float(r["gflop"]) for r in sel) * 1e9
            ex = sum(executed(r) for r in sel) * 1e9
            return ms, len(sel), fl, sum(float(r["alg_mb"]) for r in sel) * 1e6, ex

        def line(pred):
            ms_, n_, fl_, by_, ex_ = agg(pred)
            if ms_ <= 0:
                return None
            sec = ms_ * 1e-3
            mfma = all(r["kind"] != "2" for r in rows if pred(r))      # kind 2 = 1-channel-side convs: never on the MFMA
            d = {"kernel_ms_per_step": round(ms_ / nprobe, 3), "launches_per_step": n_ // nprobe,
                 "alg_GBps": round(by_ / sec / 1e9, 1), "frac_hbm": round(by_ / sec / 1e9 / PEAK_HBM_GBS, 4),
                 "tflops": round(fl_ / sec / 1e12, 2)}
            d["frac_mfma"] = round(ex_ / sec / 1e12 / peak_tf, 4) if mfma else None
            return d

        peak_tf = PEAK_FP32_MFMA_TFLOPS if args.precision == "f32" else PEAK_BF16_MFMA_TFLOPS
        # dominant kernel: fp32 -> wino_kernel (Winograd F(2x2,3x3) on the fp32 MFMA: every stride-1 3x3 conv fwd/dgrad,
        # cfg 4064); bf16 mode -> the bf16-operand patch GEMM (kind 3)
        if args.precision == "f32":
            dom, dom_name = (lambda r: r["kind"] == "0" and r["cfg"] == "4064"), \
                "wino_kernel (Winograd F(2x2,3x3) fp32-MFMA conv fwd/dgrad, 16x16 px x 64 ch tile)"
        else:
            dom, dom_name = (lambda r: r["kind"] == "3" and r["cfg"] == "4016"), \
                "wino16_kernel (Winograd F(2x2,3x3), fp32 transforms, bf16 MFMA operands, fp32 accumulate)"
        ms, n, fl, by, ex = agg(dom)
        sec = ms * 1e-3
        ach = ex / sec / 1e12 if ms > 0 else 0.0
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes (tools/pmc_traffic.sh -> profiles/rNN_pmc_traffic.json);
        # it is not measurable from inside this process, so the driver's line carries null and names the committed profile
        traffic, traffic_src = None, None
        for tname in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            if os.path.exists(os.path.join(ROOT, "profiles", tname)):
                traffic_src = "profiles/" + tname
                break
        # ... except at the workload that profile was taken on (the default: fp32, 256 x 256, batch 16, one GPU), where the
        # committed per-launch figure of the dominant kernel is repeated here (bytes; FROM THE PROFILE, not from this run)
        if traffic_src and args.precision == "f32" and args.batch == BATCH and args.size == TILE and world == 1 and not args.checkpoint:
            try:
                import json as _json
                with open(os.path.join(ROOT, traffic_src)) as _f:
                    traffic = round(float(_json.load(_f)["wino_kernel"]["hbm_bytes_per_launch"]))
            except (OSError, KeyError, ValueError):
                traffic = None
        common = {"traffic": traffic, "traffic_profile": traffic_src,
                  "effective_tflops": round(fl / sec / 1e12, 2) if ms > 0 else 0.0,
                  "launches_per_step": n // nprobe, "avg_launch_ms": round(ms / max(n, 1), 4),
                  "gflop_per_launch": round(fl / max(n, 1) / 1e9, 3),
                  "executed_gflop_per_launch": round(ex / max(n, 1) / 1e9, 3),
                  "alg_bytes_per_launch": round(by / max(n, 1)),
                  "kernel_ms_per_step": round(ms / nprobe, 3)}
        if args.precision == "f32":
            roofline = {"bound": "mfma", "kernel": dom_name, "achieved": round(ach, 2), "peak": peak_tf, "unit": "TFLOP/s",
                        "frac": round(ach / peak_tf, 4)}
        else:
            # bf16 operands: the matrix pipe is ~10 % busy by construction (one 32-cycle MFMA per transform point and
            # 16-channel step); the kernel is bound by its staging and, on the full-resolution layers, by HBM
            gbs = by / sec / 1e9 if ms > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4), "mfma_frac": round(ach / peak_tf, 4)}
        roofline.update(common)
        extra = {}
        for name, pred in [("all_conv_fwd_dgrad_mfma", lambda r: r["kind"] == "0"),
                           ("direct_pgemm_128x128_tile", lambda r: r["kind"] == "0" and r["cfg"] == "1128"),
                           ("direct_pgemm_256x64_tile", lambda r: r["kind"] == "0" and r["cfg"] == "1064"),
                           ("wgrad_mfma_all", lambda r: r["kind"] == "1"),
                           ("wgrad_winograd_3x3", lambda r: r["kind"] == "1" and r["cfg"] == "4164"),
                           ("vgg_trunk_winograd_4x4_fwd_dgrad", lambda r: r["cfg"] == "4044"),
                           ("d_4x4s2_winograd_2x2_fwd_dgrad", lambda r: r["cfg"] == "4022"),
                           ("d_4x4s2_winograd_2x2_wgrad", lambda r: r["cfg"] == "4122"),
                           ("one_channel_convs_hbm", lambda r: r["kind"] == "2"),
                           ("bf16_operand_kernels", lambda r: r["kind"] == "3")]:
            d = line(pred)
            if d is not None:
                extra[name] = d
        roofline["other_kernels"] = extra
        # the full-resolution partial-conv layers north_star's ">= 40 % of the memory roofline" is about, both ways
        layers = {}
        for lname in ("enc1", "dec1", "dec2", "final", "d2", "d5", "d8"):
            for part in ("fwd", "dgrad", "wgrad"):
                d = line(lambda r, t=f"{lname}.{part}": r.get("tag") == t)
                if d is not None:
                    layers[f"{lname}.{part}"] = d
        roofline["layers"] = layers
    if world > 1:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.size, args.batch)

    if rank == 0:
        tiles = args.batch * world * args.steps
        prec_txt = "fp32" if args.precision == "f32" else "bf16 mixed precision (bf16 MFMA operands, fp32 accumulate / storage / masters)"
        cfg_idx = {(256, 16, "f32", False): "configs[1]" if world == 1 else "configs[3] per-GPU shape",
                   (512, 8, "bf16", False): "configs[2]",
                   (1024, 4, "f32", True): "configs[4] per-GPU shape"}.get((args.size, args.batch, args.precision, args.checkpoint),
                                                                           "non-BASELINE shape")
        line = {"metric": f"DSM tiles/sec train-step (G+D) at {args.size}x{args.size} bs={args.batch}",
                "value": round(tiles / elapsed, 2),
                "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "bf16 operands / f32 accumulate+storage", "data": "synthetic",
                "config": {"workload": f"BASELINE {cfg_idx}: {args.size}x{args.size} 1-ch DSM tiles, batch {args.batch} per GPU, "
                                       f"{prec_txt}, PConv-UNet G + PatchGAN D + L1/VGG-perceptual/TV/boundary(0.5) losses + 2x Adam"
                                       + (", activation checkpointing" if args.checkpoint else ""),
                           "global_batch": args.batch * world, "tile": args.size,
                           "parallelism": f"dp{world}" if world > 1 else "single-gpu",
                           "launch": "hipGraph replay" if args.graph else "eager",
                           # what the process group itself reports (not the arguments): proof of N ranks on the named transport
                           "dist": ({"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                     "g_buckets": len(sync._plans["G"][1]) if "G" in sync._plans else None,
                                     "d_buckets": len(sync._plans["D"][1]) if "D" in sync._plans else None,
                                     "bucket_mb": round(sync.bucket_elems * 4 / (1 << 20), 1),
                                     "exchange": "all-reduce(sum) of G and D gradient buckets, each launched when its last "
                                                 "gradient is enqueued; Adam per bucket after its wait; 1/world in the Adam kernel"}
                                    if (sync is not None and dist.is_initialized()) else None),
                           "vgg_weights": "deterministic stand-in (ImageNet weights not fetchable offline; same FLOPs)",
                           "vgg_trunk": "forward Winograd F(2x2,3x3), dgrads F(4x4,3x3) (DESIGN 2b)"},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
