#!/usr/bin/env python3
"""bench.py -- CD-1 updates/s of the headline RBM (10000 <-> 1500, batch 64 per GPU) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete ``RBM.train_epoch`` call (reference rbm.py:180-227) on the global batch:
K1 -> K2 -> K1 -> K3 (+ epilogue kernels), called through the product's Python class exactly as
``iDBN.train`` calls it (idbn.py:202).  Inputs (16 distinct synthetic binary 100x100 "dot" frames
batches, density 0.1) are resident in HBM before the timed region.  With N>1 each rank holds a
full parameter replica and 64 rows of a 64*N global batch; one exchange per step over RCCL (all-gather of the
7 MB factor blocks by default, or all-reduce of the 60 MB packed statistics with --dp-mode allreduce); `value` counts batch-64 updates per second summed over the ranks (= N x global steps/s,
weak scaling: 64 rows per GPU).

One JSON line on rank 0.  `roofline` is for the dominant kernel (K3 assoc_update): algorithmic bytes
16*V*H per launch over the HIP-event-measured mean launch time on the launch stream.
`cpu_baseline` times the numpy oracle (a port of the reference arithmetic) on the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "multimodal-idbn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)



def _cpu_share() -> int:
    """CPUs this process may actually use (cgroup v2 quota; a GPU box gives one GPU's share, e.g. 16 of 256)."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(q) // int(p))
    except Exception:
        pass
    return os.cpu_count() or 1


# Thread pools sized to the visible 256 cpus overrun a 16-cpu cgroup quota: the spinning workers get the
# whole cgroup throttled and the enqueue thread stalls for tens of ms (measured: one 77 ms stall in a 40 ms
# timed region).  Size the pools to the share before torch / numpy start them.
CPU_SHARE = _cpu_share()
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, CPU_SHARE)))
os.environ.setdefault("OPENBLAS_NUM_THREADS", str(CPU_SHARE))
os.environ.setdefault("MKL_NUM_THREADS", str(CPU_SHARE))

V, H, B = 10000, 1500, 64
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA


def cpu_baseline(max_seconds: float = 12.0, max_updates: int = 60):
    """numpy oracle ("port" of rbm.py:194-227) on the identical workload; bounded sample."""
    import numpy as np
    import oracle.rbm_oracle as O
    from oracle.draws import DrawStream
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    s = DrawStream(1)
    st = O.RBMState.create((s.normal((V, H)) / np.float32(100.0)).astype(np.float32), 0.1, 1e-4, 0.5,
                           dynamic_lr=True, final_momentum=0.95)
    X = (s.uniform((B, V)) > 0.9).astype(np.float32)
    for _ in range(2):
        O.train_epoch(st, X, 0, 1, s)
    n, t0 = 0, time.perf_counter()
    while n < max_updates and time.perf_counter() - t0 < max_seconds:
        O.train_epoch(st, X, 0, 1, s)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "updates/s", "cores": int(threads), "kind": "port",
            "sample": f"{n} CD-1 updates of the same 10000x1500 batch-64 workload, numpy/OpenBLAS oracle, "
                      f"{os.cpu_count()} host cpus visible, cgroup share {CPU_SHARE}"}


def _p50_max(t0, stamps):
    d = sorted(1e6 * (b - a) for a, b in zip([t0] + stamps[:-1], stamps))
    return [round(d[len(d) // 2], 1), round(d[-1], 1)] if d else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="parity", choices=["parity", "fast"])
    ap.add_argument("--ksplit-up", type=int, default=0)
    ap.add_argument("--ksplit-down", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (tuning experiments)")
    ap.add_argument("--dp-full-planes", action="store_true", help="factor exchange: send the data plane as bf16 planes, not bits")
    ap.add_argument("--no-prefetch", action="store_true", help="do not hand train_epoch the following batch (next_data=)")
    ap.add_argument("--force-dp", action="store_true", help="take the stats/all-reduce/apply path even with one rank")
    ap.add_argument("--no-k3-events", action="store_true", help="do not bracket K3 with HIP events in the timed region")
    ap.add_argument("--dp-mode", default="factors", choices=["factors", "allreduce"],
                    help="data-parallel exchange: factor blocks (all-gather, ~7 MB/rank) or packed fp32 statistics (all-reduce, 60 MB)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank flow on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if args.backend != "nccl":
        local = local % torch.cuda.device_count()          # rehearsal: several ranks may share a device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from imdbn import engine as E
    from imdbn.engine import native
    from imdbn.models import RBM

    eng = E.get_hip_engine()
    eng.mode = native.FAST_BF16 if args.mode == "fast" else native.PARITY_F32
    if args.ksplit_up or args.ksplit_down:
        eng.set_tuning(args.ksplit_up, args.ksplit_down)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    if world > 1 or args.force_dp:
        E.dp.enable(force=args.force_dp, mode=args.dp_mode, binary_data=not args.dp_full_planes)   # the synthetic batches are 0/1 images

    torch.manual_seed(0)
    rbm = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
    if world > 1:      # identical replicas (W lives in a row-padded buffer: broadcast a contiguous copy)
        for t in (rbm.W.data, rbm.hid_bias.data, rbm.vis_bias.data):
            c = t.contiguous()
            dist.broadcast(c, 0)
            t.copy_(c)
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    batches = [(torch.rand(B, V, generator=g) > 0.9).float().to(dev) for _ in range(16)]
    E.set_rng(E.PhiloxRng(seed=2, row0=rank * B))

    def step(i):
        # a training loop knows its next batch (iDBN.train does the same lookahead): its operand forms are prepared
        # during this update's weight kernel.  Same work, same results, no separate preparation launch.
        nxt = None if args.no_prefetch else batches[(i + 1) % len(batches)]
        return rbm.train_epoch(batches[i % len(batches)], 0, 1, CD=1, next_data=nxt)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        for i in range(args.warmup):
            step(i)
        sync()
    except Exception as e:          # the factor exchange has preconditions (imdbn.engine.dp.enable); the all-reduce path has none
        if not ((world > 1 or args.force_dp) and args.dp_mode == "factors"):
            raise
        print(f"[bench] rank {rank}: factor exchange failed ({type(e).__name__}: {e}); using the all-reduce exchange", file=sys.stderr, flush=True)
        args.dp_mode = "allreduce"
        E.dp.enable(force=args.force_dp, mode="allreduce", binary_data=not args.dp_full_planes)
        for i in range(args.warmup):
            step(i)
        sync()
    if world == 1 and not args.no_k3_events:
        eng.profile(True)
    t0 = time.perf_counter()
    stamps = []
    for i in range(args.steps):
        loss = step(i)
        stamps.append(time.perf_counter())
    t_enq = time.perf_counter() - t0          # host time to enqueue all steps (== dt when host-bound)
    sync()
    dt = time.perf_counter() - t0
    k3_ms, k3_n = (eng.profile_read() if world == 1 else (0.0, 0))
    if world == 1:
        eng.profile(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(loss).item(), "loss is not finite"
    replicas_identical = None
    if world > 1:      # after the timed region: every rank applied the same update, so the replicas must agree bit for bit
        chk = torch.stack([rbm.W.data.double().sum(), rbm.W.data.double().abs().sum(), rbm.hid_bias.data.double().sum()])
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        replicas_identical = all(torch.equal(allc[0], c) for c in allc)

    if rank == 0:
        # unit = one batch-64 CD-1 update; every rank processes one per step (the step updates the shared weights
        # with the 64*N-row global batch), so the whole job processes N units per step
        ups = world * args.steps / dt
        out = {
            "metric": "CD-1 updates/sec (batch 64, 10000<->1500 RBM)",
            "value": ups, "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.mode == "parity" else "bf16",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] layer 1: RBM 10000<->1500 train_epoch, CD-1, batch 64 per GPU, "
                                   "fp32 master weights, lr 0.1 wd 1e-4 mom 0.5",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "dp_exchange": ((args.dp_mode + ("" if args.dp_mode != "factors" else (" (wire form: sample as bits)" if args.dp_full_planes else " (wire form: data and sample as bits)")))
                                       if (world > 1 or args.force_dp) else None),
                       "arithmetic": "bf16x3 split MFMA (fp32-exact products)" if args.mode == "parity" else "bf16 MFMA",
                       "final_loss": float(loss), "replicas_identical": replicas_identical},
            "global_steps_per_s": args.steps / dt,
            "host_enqueue_us_per_step": 1e6 * t_enq / args.steps,
            "host_enqueue_us_p50_max": _p50_max(t0, stamps),
            "frac_hbm_roofline_whole_step": (ups / world) * 16.0 * V * H / (HBM_PEAK_GBS * 1e9),
            "frac_bf16_mfma_roofline_whole_step": (ups / world) * 10.0 * B * V * H / (BF16_PEAK_TFLOPS * 1e12),
        }
        if world == 1 and k3_n > 0:
            avg_s = 1e-3 * k3_ms / k3_n
            ach = 16.0 * V * H / avg_s / 1e9
            traffic, tsrc = None, None        # HBM bytes per launch from the committed PMC passes of this same command
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
                for k, d in pm["kernels"].items():
                    if "assoc_update" in k:
                        traffic, tsrc = d["hbm_bytes_per_launch"], "profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc, separate passes)"
            except Exception:
                pass
            out["roofline"] = {"kernel": "assoc_update (K3)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                               "avg_launch_us": 1e6 * avg_s, "launches": k3_n,
                               "algorithmic_bytes_per_launch": 16 * V * H}
        else:
            out["roofline"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
