#!/usr/bin/env python3
"""bench.py -- DSM tiles/sec through one full GAN train step (G + D + losses + Adam).

Workload = BASELINE.json configs[1]: 256x256 1-channel synthetic DSM tiles + random disc masks, batch 16
PER GPU, fp32, partial-conv U-Net generator + PatchGAN discriminator + L1/perceptual/TV/boundary losses.
A "step" is one pass of mvp_gan.src.train.train_step over one batch that is already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = tiles processed by all ranks / max-over-ranks wall time of exactly K
steps (barrier + synchronize on both sides).  `roofline` is measured live: an extra instrumented pass after the
timed region brackets every launch of the dominant kernel (the fp32-MFMA Winograd conv) with hipEvents on
its launch stream (tg_prof_*), achieved = sum of algorithmic FLOPs / sum of kernel time.  `cpu_baseline` times
the CPU oracle (oracle/terragan_oracle.py, "port") on the host cores, rank 0 at N=1 only.
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
PEAK_HBM_GBS = 8000.0
TILE = 256
BATCH = 16


def prof_summary(lib, kind):
    ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    lib.tg_prof_summary(kind, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
    return ms.value, n.value, fl.value, by.value


def cpu_baseline(seconds_budget=25.0):
    """CPU oracle train step on the host cores: bounded sample of the same workload (batch 4, 256x256)."""
    from oracle import terragan_oracle as Orc
    # the GPU box gives a 1-GPU job a 16-core CPU share although os.cpu_count() reports the whole host
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    st = Orc.TrainState(0)
    b = 4
    real, mask = Orc.synth_batch(b, TILE, 1000)
    Orc.train_step(st, real, mask)                      # warm-up (oneDNN primitive creation)
    t0, n = time.perf_counter(), 0
    while True:
        Orc.train_step(st, real, mask)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget * 0.5 or n >= 4:
            break
    return {"value": round(b * n / el, 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} train steps of the CPU oracle at batch {b}, 256x256 fp32, after 1 warm-up step ({el:.1f} s)"}


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
    ap.add_argument("--prof-dump", default=None, help="write the per-launch table of the instrumented pass to this CSV")
    args = ap.parse_args()

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
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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

    def run(k):
        for i in range(k):
            real, mask = batches[i % nb]
            train_step(G, D, crit, oG, oD, real, mask, grad_sync=sync)

    run(args.warmup)
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

    roofline = None
    if not args.no_roofline:
        # instrumented pass (outside the timed region): hipEvents around every MFMA conv launch, on its launch stream.
        # Every rank runs the probe steps (they contain the gradient all-reduce); only rank 0 records.
        import csv
        import tempfile
        nprobe = 2
        if rank == 0:
            lib.tg_prof_enable(1)
        run(nprobe)
        torch.cuda.synchronize()
        lib.tg_prof_enable(0)
    if rank == 0 and not args.no_roofline:
        dump = args.prof_dump or os.path.join(tempfile.gettempdir(), f"tg_prof_{os.getpid()}.csv")
        lib.tg_prof_dump(dump.encode())
        rows = list(csv.DictReader(open(dump)))
        for kind in (0, 1, 2, 3):
            prof_summary(lib, kind)                       # consume the records

        def agg(pred):
            sel = [r for r in rows if pred(r)]
            ms = sum(float(r["ms"]) for r in sel)
            return ms, len(sel), sum(float(r["gflop"]) for r in sel) * 1e9, sum(float(r["alg_mb"]) for r in sel) * 1e6

        # dominant kernel symbol = wino_kernel (Winograd F(2x2,3x3) on the fp32 MFMA: every stride-1 3x3 conv fwd/dgrad):
        # cfg 4064.  `achieved` follows the contract: ALGORITHMIC flops (2*M*N*K of the direct convolution)
        # per launch / launch duration -- the kernel executes 2.25x fewer multiplies than that, so `frac` can exceed the
        # share of the MFMA pipe that is busy, which is reported next to it as `mfma_pipe_frac` (= achieved / 2.25 / peak).
        ms, n, fl, by = agg(lambda r: r["kind"] == "0" and r["cfg"] == "4064")
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):       # offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
            tj = json.load(open(tpath))
            key = [k for k in tj if k.startswith("wino_kernel")]
            traffic = round(tj[key[0]]["hbm_bytes_per_launch"]) if key else None
        roofline = {"bound": "mfma", "kernel": "wino_kernel (Winograd F(2x2,3x3) fp32-MFMA conv fwd/dgrad, 16x16 px x 64 ch tile)",
                    "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "mfma_pipe_frac": round(ach / 2.25 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "launches_per_step": n // nprobe, "avg_launch_ms": round(ms / max(n, 1), 4),
                    "gflop_per_launch": round(fl / max(n, 1) / 1e9, 3), "alg_bytes_per_launch": round(by / max(n, 1)),
                    "kernel_ms_per_step": round(ms / nprobe, 3)}
        extra = {}
        for name, pred in [("all_conv_fwd_dgrad_mfma", lambda r: r["kind"] == "0"),
                           ("direct_pgemm_128x128_tile", lambda r: r["kind"] == "0" and r["cfg"] == "1128"),
                           ("direct_pgemm_256x64_tile", lambda r: r["kind"] == "0" and r["cfg"] == "1064"),
                           ("wgrad_mfma_all", lambda r: r["kind"] == "1"),
                           ("wgrad_winograd_3x3", lambda r: r["kind"] == "1" and r["cfg"] == "4164"),
                           ("one_channel_convs_hbm", lambda r: r["kind"] == "2"),
                           ("pgemm_bf16_operands", lambda r: r["kind"] == "3")]:
            ms_, n_, fl_, by_ = agg(pred)
            if ms_ > 0:
                extra[name] = {"kernel_ms_per_step": round(ms_ / nprobe, 3), "launches_per_step": n_ // nprobe,
                               "TFLOPs": round(fl_ / (ms_ * 1e-3) / 1e12, 2), "alg_GBps": round(by_ / (ms_ * 1e-3) / 1e9, 1)}
        roofline["other_kernels"] = extra
    if world > 1:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        tiles = args.batch * world * args.steps
        line = {"metric": "DSM tiles/sec train-step (G+D) at 256x256 bs=16", "value": round(tiles / elapsed, 2),
                "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "bf16 operands / f32 accumulate+storage", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[1]: {args.size}x{args.size} 1-ch DSM tiles, batch {args.batch} per GPU, "
                                       "fp32, PConv-UNet G + PatchGAN D + L1/VGG-perceptual/TV/boundary losses + 2x Adam",
                           "global_batch": args.batch * world, "tile": args.size,
                           "parallelism": f"dp{world}" if world > 1 else "single-gpu",
                           "vgg_weights": "deterministic stand-in (ImageNet weights not fetchable offline; same FLOPs)"},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
