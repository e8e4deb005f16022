#!/usr/bin/env python3
"""bench.py -- CD-1 updates/s of the headline RBM (10000 <-> 1500, batch 64 per GPU) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work: with ``--gpus N > 1`` and no torchrun environment the process starts N fresh rank
processes itself (``python -m torch.distributed.run`` as a CHILD, before anything here has touched the
GPU) and relays rank 0's JSON line.

A "step" is one complete ``RBM.train_epoch`` call (reference rbm.py:180-227) on the global batch:
K1 -> K2 -> K1 -> K3, called through the product's Python class exactly as ``iDBN.train`` calls it
(idbn.py:202).  Inputs (16 distinct synthetic binary 100x100 "dot" frame batches, density 0.1) are resident
in HBM before the timed region.  With N>1 each rank holds a full parameter replica and 64 rows of a 64*N
global batch; one exchange per step over RCCL (all-gather of the factor blocks by default, the all-reduce of
the packed statistics that north_star names is timed right after it and reported beside it); `value` counts
batch-64 updates per second summed over the ranks (= N x global steps/s, weak scaling: 64 rows per GPU).

One JSON line on rank 0.  `roofline` is for the dominant kernel (K3 assoc_update): algorithmic bytes
16*V*H per launch over the HIP-event-measured mean launch time on the launch stream.
`cpu_baseline` times the numpy oracle (a port of the reference arithmetic) on the same workload.
`other_configs` times the other BASELINE.json configs (C2 stack, C3, C5) after the headline region.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
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
TRAFFIC_JSON = os.path.join("profiles", "r03_pmc_hbm_traffic.json")
ROCPROF_STATS_CSV = os.path.join("profiles", "r03_bench_kernel_stats.csv")


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _engine_source_sha() -> str:
    """Hash of the kernel sources: stamps measurements that were taken in another run (PMC traffic)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "multimodal-idbn_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(max_seconds: float = 12.0, max_updates: int = 60):
    """numpy oracle ("port" of rbm.py:194-227) on the identical workload; bounded sample."""
    import numpy as np
    import oracle.rbm_oracle as O
    from oracle.draws import DrawStream
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = int(os.environ.get("OPENBLAS_NUM_THREADS", "1"))
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
    return {"value": n / dt, "unit": "updates/s", "cores": int(threads), "kind": "port", "cpu_model": _cpu_model(),
            "ms_per_update": 1e3 * dt / max(n, 1),
            "sample": f"{n} CD-1 updates of the same 10000x1500 batch-64 workload, numpy/OpenBLAS oracle with {int(threads)} BLAS threads, "
                      f"{os.cpu_count()} host cpus visible, cgroup share {CPU_SHARE}"}


def other_configs(dev):
    """The other BASELINE.json configs through the product classes (1 GPU, synthetic data), after the headline region.
    Bounds: the 1500<->500 layer, the joint RBM and the chains are launch / dependency-latency bound (weights of 3 MB and
    0.5 MB sit in L2); C5's decode streams the 63 MB of image-stack weights once (HBM bound: 63 MB / t against 8 TB/s)."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from imdbn import engine as E
    from imdbn.models import RBM, iDBN, iMDBN

    E.manual_seed(3)
    params = {"LEARNING_RATE": 0.1, "WEIGHT_PENALTY": 1e-4, "INIT_MOMENTUM": 0.5, "FINAL_MOMENTUM": 0.95,
              "LEARNING_RATE_DYNAMIC": True, "CD": 1, "JOINT_LEARNING_RATE": 0.04, "JOINT_CD": 1,
              "JOINT_AUX_COND_STEPS": 30, "CROSS_GIBBS_STEPS": 50}
    out = {}

    def timeit(fn, n, warm=2, warm_s=0.0):
        """mean wall time of fn() over n calls after `warm` calls and at least `warm_s` seconds of them (the first 100 ms of a new mix
        of operations carry one-off stalls of 30-50 ms -- first use of kernels that are loaded lazily: two of them fell on calls 0
        and 2 of the live best-of-16 loop, i.e. one inside a timed region of 10 calls)"""
        t_w = time.perf_counter()
        i = 0
        while i < warm or time.perf_counter() - t_w < warm_s:
            fn()
            i += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    cwd = os.getcwd()
    import tempfile
    with tempfile.TemporaryDirectory() as scratch:      # iDBN.__init__ creates logs-idbn/ in cwd (reference idbn.py:115-116)
        os.chdir(scratch)
        try:
            X = (torch.rand(64 * 8, 10000) > 0.9).float()
            dl = DataLoader(TensorDataset(X, torch.zeros(len(X), 1)), batch_size=64)
            # C2: the product loop itself (iDBN.train, idbn.py:195-204) over a device-resident sequential loader of 32 batches
            X2 = (torch.rand(64 * 32, 10000, device=dev) > 0.9).float()
            dl2 = DataLoader(TensorDataset(X2, torch.zeros(len(X2), 1, device=dev)), batch_size=64, shuffle=False)
            d = iDBN([10000, 1500, 500], dict(params), dl2, dl2, dev)
            d.train(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d.train(3)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / (3 * len(dl2))
            # algorithmic bytes of one batch through both layers: a CD-1 update streams W three times and
            # read-modify-writes W and W_m once (7 x 4VH bytes), the forward for the next layer reads W once more
            c2_bytes = 8 * 4 * 10000 * 1500 + 7 * 4 * 1500 * 500
            out["C2_stack"] = {"ms_per_batch": 1e3 * t, "batches_per_s": 1.0 / t,
                               "bound": "layer 1 HBM (see roofline); layer 2 + forwards launch/latency",
                               "ref_cpu_ms_per_batch": 213.0,
                               "roofline": {"bound": "hbm", "achieved": c2_bytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": c2_bytes / t / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                            "note": "bytes = 8 x 4VH (layer 1: three reads + W, W_m read-modify-write + the forward) + 7 x 4VH (layer 2)"}}

            jr = RBM(532, 256, 0.04, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95, softmax_groups=[(500, 532)]).to(dev)
            z = torch.rand(64, 500, device=dev)
            y = torch.eye(32, device=dev)[torch.randint(0, 32, (64,), device=dev)]
            vp = torch.cat([z, y], 1)
            vk = torch.zeros(64, 532, device=dev)
            km = torch.zeros(64, 532, device=dev)
            vk[:, 500:] = y
            km[:, 500:] = 1

            def c3_main():      # imdbn.py:590-611: one joint CD-1 update + one auxiliary clamped update (30 init steps)
                jr.train_epoch(vp, 9, 20, CD=1)
                jr.train_epoch_clamped(vk, km, 9, 20, CD=1, cond_init_steps=30, sample_h=False, sample_v=False,
                                       reclamp_negative=False, aux_lr_mult=0.3, use_noisy_init=True)

            def c3_warm():      # imdbn.py:566-588: two clamped updates
                for _ in range(2):
                    jr.train_epoch_clamped(vk, km, 0, 20, CD=1, cond_init_steps=30, sample_h=False, sample_v=False,
                                           aux_lr_mult=0.3, use_noisy_init=True)
            t3 = timeit(c3_main, 20)
            out["C3_joint_532x256"] = {"main_step_ms": 1e3 * t3, "warmup_step_ms": 1e3 * timeit(c3_warm, 20),
                                       "bound": "dependency latency (61 half steps on a 545 KB matrix)", "ref_cpu_main_ms": 30.7,
                                       # no bandwidth or MFMA roofline binds a 545 KB matrix: the unit is the dependent half step
                                       "roofline": {"bound": "latency", "achieved": 1e6 * t3 / 66.0, "peak": None, "unit": "us per dependent half step",
                                                    "frac": None, "traffic": None,
                                                    "note": "66 dependent half steps per main step (CD-1 update: 3 + its update; clamped update: 2 x 30 init + 3); "
                                                            "the chain kernel runs them in one launch (DESIGN.md 9.4)"}}

            m = iMDBN([10000, 1500, 500], 256, params=dict(params), dataloader=dl, val_loader=dl, device=dev, num_labels=32)
            m.z_class_mean = torch.rand(32, 500, device=dev)
            z5 = torch.rand(256, 500, device=dev)
            y5 = torch.eye(32, device=dev)[torch.randint(0, 32, (256,), device=dev)]
            t5 = timeit(lambda: m._cross_reconstruct(z5, y5, steps=50), 10, warm_s=0.3)
            m.live_best_of_k, m.best_of_k = True, 16      # K=16 with live free-energy selection
            t5k = timeit(lambda: m._cross_reconstruct(z5, y5, steps=50), 10, warm_s=0.3)
            td = timeit(lambda: m.image_idbn.decode(z5), 10)
            xr = (torch.rand(256, 10000, device=dev) > 0.9).float()
            tr = timeit(lambda: m.image_idbn.represent(xr), 10)
            # decode / represent of 256 real-valued rows: 9 bf16 products per fp32 product (3 x 3 terms), dense bf16 MFMA peak
            dec_flops = 2.0 * 256 * (500 * 1500 + 1500 * 10000) * 9
            out["C5_cross_reconstruct_b256_s50"] = {"ms_default_k5_inert": 1e3 * t5, "ms_live_k16": 1e3 * t5k, "decode_ms": 1e3 * td,
                                                    "represent_ms": 1e3 * tr,
                                                    "bound": "chains: dependency latency; decode / represent: MFMA (9 bf16 products per fp32 product)",
                                                    "ref_cpu_ms": 263.0,
                                                    "roofline": {"bound": "latency", "achieved": 1e6 * (t5 - td) / 100.0, "peak": None,
                                                                 "unit": "us per dependent chain step", "frac": None, "traffic": None,
                                                                 "note": "2 chains x 50 steps in one launch of the chain kernel, then the decode"},
                                                    "decode_roofline": {"bound": "mfma", "achieved": dec_flops / td / 1e12, "peak": BF16_PEAK_TFLOPS,
                                                                        "unit": "TFLOP/s", "frac": dec_flops / td / 1e12 / BF16_PEAK_TFLOPS, "traffic": None,
                                                                        "note": "bf16 MFMA flops issued = 9 x 2 x 256 x (500x1500 + 1500x10000); wall time of "
                                                                                "iDBN.decode (4 launches)"}}

            # iMDBN.train_joint (imdbn.py:540-640): the C3 updates + the per-batch cross-modal metrics, overlapped on a second stream
            # (32 batches per epoch, 4 timed epochs: every train_joint call starts with init_joint_bias_from_data over 10 batches and
            #  ends each epoch with one host read of the metrics -- 20-30 ms per call, which a loop of 32 batches would not amortise)
            yj = torch.randint(0, 32, (len(X2),))
            dlj = DataLoader(TensorDataset(X2, torch.eye(32)[yj].to(dev)), batch_size=64, shuffle=False)
            import contextlib, io
            tj = {}
            for ov in (True, False):
                mj = iMDBN([10000, 1500, 500], 256, params=dict(params, JOINT_METRICS_OVERLAP=ov), dataloader=dlj, val_loader=dlj, device=dev, num_labels=32)
                with contextlib.redirect_stdout(io.StringIO()):
                    mj.train_joint(1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    mj.train_joint(4)
                    torch.cuda.synchronize()
                tj[ov] = (time.perf_counter() - t0) / (4 * len(dlj))
            out["C3_train_joint_loop"] = {"ms_per_batch": 1e3 * tj[True], "ms_per_batch_metrics_inline": 1e3 * tj[False],
                                          "bound": "dependency latency; the metrics' chains run on a second stream against a snapshot of the joint RBM",
                                          # per batch on the training stream: represent (2 launches) + a CD-1 update + a clamped update with 30 init
                                          # steps = 66 dependent half steps (C3 main step); the metrics (2 x 50 chain steps + decode) ride beside them
                                          "roofline": {"bound": "latency", "achieved": 1e6 * tj[True] / 66.0, "peak": None,
                                                       "unit": "us per dependent half step of the training stream", "frac": None, "traffic": None,
                                                       "note": "with the metrics in line the batch has 66 + 200 dependent half steps"}}
        finally:
            os.chdir(cwd)
    return out


def _p50_max(t0, stamps):
    d = sorted(1e6 * (b - a) for a, b in zip([t0] + stamps[:-1], stamps))
    return [round(d[len(d) // 2], 1), round(d[-1], 1)] if d else None


def _spawn_ranks(n: int, argv) -> int:
    """Parent of a multi-GPU run started as plain `python bench.py --gpus N`: start the ranks as a child process tree
    (this process has not imported torch or touched the GPU) and pass their output through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import __graft_entry__ as ge
    ge.build()                      # once, here, before any rank exists
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print("[bench] starting ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # this pool's driver only has dmabuf IPC (RCCL needs it across processes)
    return subprocess.call(cmd, env=env)


def _under_profiler() -> bool:
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_ROOT", "HSA_TOOLS_LIB"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C2 / C3 / C5 timings after the headline region")
    ap.add_argument("--mode", default="parity", choices=["parity", "fast"])
    ap.add_argument("--ksplit-up", type=int, default=0)
    ap.add_argument("--ksplit-down", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (tuning experiments)")
    ap.add_argument("--dp-full-planes", action="store_true", help="factor exchange: send the data plane as bf16 planes, not bits")
    ap.add_argument("--no-prefetch", action="store_true", help="do not hand train_epoch the following batch (next_data=)")
    ap.add_argument("--force-dp", action="store_true", help="take the data-parallel path even with one rank")
    ap.add_argument("--no-k3-events", action="store_true", help="do not bracket K3 with HIP events in the timed region")
    ap.add_argument("--dp-mode", default="factors", choices=["factors", "allreduce"],
                    help="data-parallel exchange of the headline region: factor blocks (all-gather) or packed fp32 statistics "
                         "(all-reduce, 60 MB); the other one is timed after it and reported as dp_other_exchange")
    ap.add_argument("--tag-batches", action="store_true",
                    help="diagnostic: mark the resident batches as 0/1 (what a DeviceLoader does); by default they are plain, untagged "
                         "tensors and the device finds out what they contain")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank flow on one GPU)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (before anything initialises the GPU; see _spawn_ranks)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # The native library is built (or found fresh) BEFORE this process initialises the GPU or a profiler's preloaded
    # library does: hipcc is never started from a GPU-initialised process.  Under a profiler, and on non-zero ranks,
    # a stale library is an error / is waited for, never compiled here.
    import __graft_entry__ as ge
    if local == 0:
        ge.build(compile_ok=not _under_profiler())
    else:
        ge.wait_built()

    import torch
    import torch.distributed as dist

    if args.backend != "nccl":
        local = local % torch.cuda.device_count()          # rehearsal: several ranks may share a device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dp = world > 1 or args.force_dp
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

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

    torch.manual_seed(0)
    rbm = RBM(V, H, 0.1, 1e-4, 0.5, dynamic_lr=True, final_momentum=0.95).to(dev)
    if world > 1:      # identical replicas (W lives in a row-padded buffer: broadcast a contiguous copy)
        for t in (rbm.W.data, rbm.hid_bias.data, rbm.vis_bias.data):
            c = t.contiguous()
            dist.broadcast(c, 0)
            t.copy_(c)
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    batches = [(torch.rand(B, V, generator=g) > 0.9).float().to(dev) for _ in range(16)]
    if args.tag_batches:
        for b in batches:
            b._imdbn_binary = True
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

    def timed_region(k3_events: bool):
        """W warm-up steps, then exactly K steps between barrier + synchronize pairs; max over the ranks."""
        if k3_events:
            eng.profile(True)      # (creates the event pool on first use: before the warm-up, so that the GPU does not sit idle behind it)
        for i in range(args.warmup):
            step(i)
        sync()
        i0 = args.warmup
        if k3_events:
            eng.profile(True)      # resets the counters: only launches of the timed region are bracketed
        t0 = time.perf_counter()
        stamps = []
        for i in range(i0, i0 + args.steps):      # continues the batch sequence of the warm-up steps
            loss = step(i)
            stamps.append(time.perf_counter())
        t_enq = time.perf_counter() - t0          # host time to enqueue all steps (== dt when host-bound)
        sync()
        dt = time.perf_counter() - t0
        k3 = eng.profile_read() if k3_events else (0.0, 0)
        if k3_events:
            eng.profile(False)
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert torch.isfinite(loss).item(), "loss is not finite"
        return dt, t_enq, stamps, t0, k3, float(loss)

    # Any failure below (an engine error, a HIP or RCCL error) ends the run with a non-zero exit code: no fallback to another
    # exchange in a process that has seen a GPU error, and no rank ever switches collectives on its own.
    if use_dp:
        E.dp.enable(force=args.force_dp, mode=args.dp_mode, binary_data=not args.dp_full_planes)   # the synthetic batches are 0/1 images
    # set-up, before the W warm-up steps: one pass over the resident batches (workspace allocation, one-time kernel attributes,
    # first touch of every batch and of both prefetch slots), so that whatever --steps / --warmup the caller picks, no first-time
    # work of any kind sits in the timed region.  The weights it leaves behind are the starting point of the measured run.
    # 16 passes (256 updates, ~26 ms): a device coming out of the build / import idle runs its first ~20 ms of work 3-4 % slower than
    # its steady state (measured: 20 timed steps after 16 set-up updates 104.9 us per step, after 256 101.0-102.4, after 1024 102.1;
    # 200 timed steps 98.6-100.8) -- `value` is a throughput, so the timed region starts on a device that is already up.
    n_prime = int(os.environ.get("IMDBN_BENCH_PRIME", 16 * len(batches)))
    for i in range(n_prime):
        step(i)
    sync()
    dt, t_enq, stamps, t0, (k3_ms, k3_n), loss = timed_region(not args.no_k3_events and (not use_dp or args.dp_mode == "factors"))

    def replicas_identical():
        if world == 1:
            return None      # every rank applied the same update, so the replicas must agree bit for bit
        chk = torch.stack([rbm.W.data.double().sum(), rbm.W.data.double().abs().sum(), rbm.hid_bias.data.double().sum()])
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        return all(torch.equal(allc[0], c) for c in allc)

    def exchange_name(mode):
        if mode != "factors":
            return "allreduce (packed fp32 statistics, one all-reduce per step)"
        return "factors (all-gather; wire form: " + ("sample as bits)" if args.dp_full_planes else "data and sample as bits)")

    ident = replicas_identical()
    other_exchange = None
    if use_dp:      # the other exchange, same K steps, reported beside the headline one
        other = "allreduce" if args.dp_mode == "factors" else "factors"
        E.dp.enable(force=args.force_dp, mode=other, binary_data=not args.dp_full_planes)
        dt2, _, _, _, _, loss2 = timed_region(False)
        other_exchange = {"dp_exchange": exchange_name(other), "value": world * args.steps / dt2, "unit": "updates/s",
                          "ms_per_step": 1e3 * dt2 / args.steps, "final_loss": loss2, "replicas_identical": replicas_identical()}

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
                       "dp_exchange": exchange_name(args.dp_mode) if use_dp else None,
                       "arithmetic": "bf16x3 split MFMA (fp32-exact products)" if args.mode == "parity" else "bf16 MFMA",
                       "final_loss": loss, "replicas_identical": ident},
            "global_steps_per_s": args.steps / dt,
            "setup_steps_before_warmup": n_prime,
            "host_enqueue_us_per_step": 1e6 * t_enq / args.steps,
            "host_enqueue_us_p50_max": _p50_max(t0, stamps),
            "frac_hbm_roofline_whole_step": (ups / world) * 16.0 * V * H / (HBM_PEAK_GBS * 1e9),
            "frac_bf16_mfma_roofline_whole_step": (ups / world) * 10.0 * B * V * H / (BF16_PEAK_TFLOPS * 1e12),
            "dp_other_exchange": other_exchange,
        }
        if k3_n > 0:
            avg_s = 1e-3 * k3_ms / k3_n
            ach = 16.0 * V * H / avg_s / 1e9
            # HBM bytes per launch come from rocprofv3 --pmc passes of this same command (separate runs: counters cannot be
            # collected from inside the process); the file carries the hash of the kernel sources it was measured on
            traffic, tsrc, stale = None, None, None
            try:
                pm = json.load(open(os.path.join(ROOT, TRAFFIC_JSON)))
                for k, d in pm["kernels"].items():
                    if "assoc_update" in k:
                        traffic, tsrc = d["hbm_bytes_per_launch"], TRAFFIC_JSON + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
                stale = pm.get("engine_source_sha") != _engine_source_sha()
            except Exception:
                pass
            kname = "assoc_update (K3)"
            if use_dp:       # the update from all ranks' factor blocks in one launch: the same 240 MB of weights move once
                kname = f"assoc_update_planes_ranks (K3, {world} rank block{'s' if world > 1 else ''})" if world > 1 else "assoc_update (K3, data-parallel path)"
                traffic, tsrc, stale = None, None, None
            out["roofline"] = {"kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                               "traffic_measured_on_other_sources": stale,
                               "avg_launch_us": 1e6 * avg_s, "launches": k3_n,
                               "algorithmic_bytes_per_launch": 16 * V * H}
            # the committed rocprofv3 --kernel-trace --stats summary of this command, beside the live event brackets (a bracket
            # = the launch between two hipEventRecords: it includes the dispatch gap on either side of the kernel, ~3 us)
            if not use_dp:
                try:
                    import csv
                    for r in csv.DictReader(open(os.path.join(ROOT, ROCPROF_STATS_CSV))):
                        if "assoc_update_planes" in r["Name"]:
                            us = float(r["AverageNs"]) / 1e3
                            out["roofline"].update({"rocprofv3_avg_launch_us": us, "rocprofv3_frac": 16 * V * H / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                    "rocprofv3_source": ROCPROF_STATS_CSV + (" (other kernel sources)" if stale else "")})
                            break
                except Exception:
                    pass
        else:
            out["roofline"] = None
        out["other_configs"] = other_configs(dev) if (world == 1 and not use_dp and not args.no_other_configs) else None
        out["cpu_baseline"] = cpu_baseline() if (world == 1 and not args.no_cpu_baseline) else None
        print(json.dumps(out), flush=True)
    if use_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
