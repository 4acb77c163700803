"""Benchmark: TT-cores sketched / second (fp64) of the streaming TT sketch on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3|c2|c4|c5] [--scaling weak|strong]

Default workload (BASELINE.json configs[2] / SURVEY.md 8d "C3", the configuration the metric is quoted on):
TensorTrain d=6, n=200, TT-rank 100, TensorTrainDRM left rank 50 (r_out) / right rank 100, all fp64, synthetic
Gaussian cores generated on the device.  A step sketches `--batch` such TTs per GPU in one batched pass
(ttsk_tt_sketch_batch: both chains, Omega, Psi) with inputs and DRMs resident in HBM.

  --scaling weak    (default) every rank sketches its own batch per step; with N > 1 the rank's partial
                    sketches are summed and ONE RCCL all-reduce per step gives every rank the sketch of the
                    whole N x batch term sum.  value = cores sketched by all ranks per second.
  --scaling strong  a fixed job of `--items` TTs (default 128) is dealt over the ranks (shard_bounds); a step
                    sketches the whole job: items / N per rank in batched passes, local sum, ONE all-reduce.
                    At N = 1 this is the weak workload repeated items / batch times per step.
  --config c2|c4|c5 the other BASELINE configurations on one GPU (c5 also shards its 32 terms over N ranks),
                    with the roofline SURVEY.md 8d names for each.

Ranks are launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; the launcher
only sets RANK / LOCAL_RANK / WORLD_SIZE.  This process never imports torch: the RCCL id travels through
tt_sketch_amd.rendezvous (files), barriers and the max-over-ranks clock are RCCL calls.

The JSON line also carries
  roofline      the dominant product of the pipeline: algorithmic flops / hipEvent time per launch (bracketing
                the fused chain-step kernel AND its slab reduce) against the fp64 MFMA peak
  cpu_baseline  the CPU oracle (NumPy restatement of the reference, same einsum calls) on the same inputs, on
                this box's host cores: BLAS thread sweep, T_sketch and T_total (incl. DRM sampling)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, N_MODE, S_IN, L_RANK, R_RANK = 6, 200, 100, 50, 100
HBM_TBS = 8.0            # TB/s, MI355X_MICROARCH.md (achievable stream rates measured here: 4.6-5.4)
PEAK_F64_MFMA_TF = 78.6   # AMD MI355X data sheet (fp64 matrix); profiles/scripts/mfma_probe2.hip measures 77.7 at 2.4 GHz
METRIC = "TT-cores sketched/sec (fp64) + achieved MFMA % for d=6 n=200 r=50 stream_sketch"


def algorithmic_flops(shape, s, l, r):
    """cheapest pairwise order of every 3-operand einsum, unfused."""
    d = len(shape)
    S = (1,) + tuple(s) + (1,)
    Lr = (1,) + tuple(l)
    Rr = tuple(r) + (1,)

    def chain(rk, S, shape):
        tot = 2 * shape[0] * S[1] * rk[1]
        for mu in range(1, d - 1):
            a, n, b, p, q = S[mu], shape[mu], S[mu + 1], rk[mu], rk[mu + 1]
            tot += min(2 * a * p * n * b + 2 * p * n * b * q, 2 * a * p * n * q + 2 * a * n * b * q)
        return tot
    left = chain(Lr, S, shape)
    right = chain((1,) + tuple(r[::-1]), S[::-1], shape[::-1])     # the right chain walks the transposed tensor
    omega = sum(2 * l[mu] * S[mu + 1] * r[mu] for mu in range(d - 1))
    psi = 2 * shape[0] * S[1] * Rr[0] + 2 * Lr[d - 1] * S[d - 1] * shape[d - 1]
    for mu in range(1, d - 1):
        a, n, b, p, q = S[mu], shape[mu], S[mu + 1], Lr[mu], Rr[mu]
        psi += min(2 * p * a * n * b + 2 * p * n * b * q, 2 * a * n * b * q + 2 * p * a * n * q)
    # what the one-call path EXECUTES: Psi_mu = T_mu R_mu reuses T_mu = L_{mu-1}^T X_mu of the left chain (the reference
    # recomputes it, tensor_train_sketch.py:28-34): the algorithmic count minus that first product of every interior Psi
    shared = sum(2 * Lr[mu] * S[mu] * shape[mu] * S[mu + 1] for mu in range(1, d - 1))
    return dict(left=left, right=right, omega=omega, psi=psi, total=left + right + omega + psi,
                executed=left + right + omega + psi - shared)


def load_traffic(kernel_name):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/r02_traffic.json,
    written by profiles/collect_traffic.py: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide
    streaming reads on gfx950, plus WRITE_SIZE)."""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                table = json.load(f)
            for key, entry in table.items():
                if isinstance(entry, dict) and (key == kernel_name or key.startswith(kernel_name.split("<")[0]) and kernel_name in key):
                    return entry.get("bytes_per_launch")
    return None


def metric_name():
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f).get("metric", METRIC)
    except (OSError, ValueError):
        return METRIC


# --------------------------------------------------------------------------- CPU baseline (the oracle = "port")
def cpu_baseline_tt(shape, cores, lcores, rcores, l_rank, r_rank, budget_s=20.0):
    """Oracle (same einsum strings / optimize flags as the reference) on host cores: BLAS thread sweep for
    T_sketch (DRMs pre-built, = general_sketch) and T_total (incl. DRM sampling, the reference's own timed
    region scripts/experiment_base.py:102-113) at the best thread count."""
    import __graft_entry__ as ge
    ge.build_oracle()
    from oracle import ttsk_oracle as orc
    d = len(shape)
    ld, rd = orc.TTDrm(lcores, shape, False), orc.TTDrm(rcores, shape, True)
    res = orc.general_sketch("tt", cores, ld, rd, "streaming")       # warm-up, and the parity reference
    import contextlib
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        maxthr = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        limit = lambda t: threadpool_limits(limits=t)
    except Exception:
        maxthr = os.cpu_count() or 1
        limit = lambda t: contextlib.nullcontext()
    sweep = {}
    counts = sorted({t for t in (8, 16, 32, 64, maxthr) if t <= maxthr})
    per = budget_s * 0.6 / max(len(counts), 1)
    for t in counts:
        with limit(t):
            times, t_end = [], time.perf_counter() + per
            while len(times) < 2 or (time.perf_counter() < t_end and len(times) < 12):
                t0 = time.perf_counter()
                orc.general_sketch("tt", cores, ld, rd, "streaming")
                times.append(time.perf_counter() - t0)
            sweep[t] = min(times)
    best_t = min(sweep, key=sweep.get)
    with limit(best_t):
        rng = np.random.default_rng(0)
        tot = []
        for _ in range(2):
            t0 = time.perf_counter()
            l2, r2 = orc.random_tt_drm(shape, l_rank, False, rng), orc.random_tt_drm(shape, r_rank, True, rng)
            orc.general_sketch("tt", cores, l2, r2, "streaming")
            tot.append(time.perf_counter() - t0)
    best = sweep[best_t]
    return dict(value=d / best, unit="TT-cores/s", cores=int(best_t), kind="port",
                t_sketch_ms=best * 1e3, t_total_ms_incl_drm_sampling=min(tot) * 1e3,
                thread_sweep_ms={str(k): round(v * 1e3, 1) for k, v in sweep.items()},
                sample=f"general_sketch of ONE d={d} n={shape[0]} TT (DRMs pre-built) repeated per BLAS thread count "
                       f"{counts}; best of each; T_total = DRM sampling (single NumPy stream) + sketch, best of 2"), res


# --------------------------------------------------------------------------- helpers
CPU_THREADS = 16     # BLAS threads of the CPU legs: the sweep of cpu_baseline_tt finds 8-16 best on the GPU boxes' hosts
                     # (their 64+ hardware threads oversubscribe these skinny products: 2-4 x slower)


def blas_threads(n=CPU_THREADS):
    import contextlib
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n)
    except Exception:
        return contextlib.nullcontext()


def cpu_cores_used():
    return min(CPU_THREADS, os.cpu_count() or 1)


def device_tt(shape, rank, seed):
    """Random TT generated in HBM: cores N(0,1)/sqrt(r1*n) (SURVEY 8d inputs)."""
    from tt_sketch_amd import TensorTrain
    from tt_sketch_amd.utils import random_normal_dev
    S = (1,) + (tuple(rank) if not np.isscalar(rank) else (rank,) * (len(shape) - 1)) + (1,)
    return TensorTrain([random_normal_dev((S[k], shape[k], S[k + 1]), seed=(seed << 8) + k, scale=1.0 / np.sqrt(S[k] * shape[k]))
                        for k in range(len(shape))])


def host_cores(tt):
    return [np.asarray(c) for c in tt.cores]


class Job:
    """Clock, barrier and reporting shared by the configurations."""

    def __init__(self, args):
        from tt_sketch_amd import _native as nat
        self.nat = nat
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            if self.world == 1 and args.gpus > 1:
                raise SystemExit("launch with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` for N > 1")
            args.gpus = self.world
        self.comm = None
        force = bool(os.environ.get("TTSK_BENCH_FORCE_COMM"))      # rehearsal: the collective path with one rank
        dev = os.environ.get("TTSK_BENCH_DEVICE")                  # rehearsal override on a one-GPU box
        if self.world > 1 or force:
            from tt_sketch_amd.distributed import RcclComm
            self.comm = RcclComm.from_env(device=None if dev is None else int(dev))
        else:
            nat.call("ttsk_init", int(dev) if dev is not None else int(os.environ.get("LOCAL_RANK", "0")))

    def barrier(self):
        if self.comm is not None:
            self.comm.barrier()
        else:
            self.nat.call("ttsk_sync", -1)

    def timed(self, step, steps, warmup):
        for _ in range(warmup):
            step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.comm is not None:
            elapsed = self.comm.max_over_ranks(elapsed)
        return elapsed

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def prof_classes(nat, reps, labels):
    classes = {}
    for c, label in labels.items():
        n_l, ms, flops = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        nat.call("ttsk_prof_read", c, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(flops))
        kname = ctypes.create_string_buffer(96)
        nat.call("ttsk_prof_kernel_name", c, kname, 96)
        if n_l.value:
            classes[label] = dict(kernel=kname.value.decode() if c != 5 else "small_gemm_kernel / gemm_f64_kernel<...> (several)",
                                  launches_per_step=n_l.value / reps, avg_us=1e3 * ms.value / n_l.value,
                                  gflop_per_launch=flops.value / n_l.value * 1e-9,
                                  tflops=flops.value / (ms.value * 1e-3) * 1e-12 if ms.value else 0.0,
                                  share_ms=ms.value / reps)
    return classes


# --------------------------------------------------------------------------- C3 (the headline) and its scaling modes
def bench_c3(args, job):
    nat = job.nat
    from tt_sketch_amd import TensorTrainDRM
    from tt_sketch_amd.device import DevArray
    from tt_sketch_amd.distributed import shard_bounds
    from tt_sketch_amd.tt_fused import TTSketchPlan
    rank, world = job.rank, job.world
    shape = (N_MODE,) * D
    B = max(1, int(args.batch))
    strong = args.scaling == "strong"
    if strong:
        lo, hi = shard_bounds(int(args.items), rank, world)
        n_mine = hi - lo
        seeds = list(range(1000 + lo, 1000 + hi))
    else:
        n_mine = B
        seeds = [1000 + rank * B + b for b in range(B)]
    tts = [device_tt(shape, S_IN, s) for s in seeds]
    left = TensorTrainDRM(L_RANK, shape, False, seed=1)            # device-sampled: identical on every rank by seed
    right = TensorTrainDRM(R_RANK, shape, True, seed=2)
    plan = TTSketchPlan(shape, (S_IN,) * (D - 1), left, right)
    stride = plan.size + (plan.size & 1)            # even spacing keeps every sketch 16-byte aligned
    passes = [(b0, min(B, n_mine - b0)) for b0 in range(0, n_mine, B)]
    inflight = max(1, min(int(args.inflight), nat.NUM_STREAMS // 2))
    comm_on = job.comm is not None
    # (collective path, strong too: two slots, so that the collective of step s travels under the products of step s + 1)
    nslots = inflight if (not strong or comm_on) else 1
    outs = [DevArray.empty((max(n_mine, 1) * stride,)) for _ in range(nslots if not strong else 1)]
    sums = [DevArray.empty((plan.size,)) for _ in range(nslots)] if comm_on else None
    keep, ptr_sets = [], []
    for b0, cnt in passes:
        flat = []
        for t in tts[b0:b0 + cnt]:
            p1, k1 = plan.core_pointers(t)
            keep.append(k1)
            flat += [p1[i] for i in range(plan.d)]
        ptr_sets.append((ctypes.c_void_p * len(flat))(*flat))
    counter = [0]
    cs = nat.NUM_STREAMS - 1                        # the one stream every collective goes to, in step order
    P = ctypes.c_void_p

    use_reduce = getattr(args, "collective", "reduce") == "reduce"

    def allreduce_slot(slot, first_stream):
        """ONE collective of the rank's partial sketch (already summed over its TTs by ttsk_tt_sketch_sum): a REDUCE to rank 0
        -- the rank that would assemble (to_tt); north_star's "single RCCL reduce", half the link traffic -- or, --collective
        allreduce, the sum on every rank"""
        nat.call("ttsk_stream_wait", cs, first_stream)
        if n_mine == 0:
            nat.call("ttsk_memset", P(sums[slot].ptr), 0, ctypes.c_size_t(plan.size * 8), cs)
        if use_reduce:
            nat.call("ttsk_comm_reduce_sum", P(sums[slot].ptr), ctypes.c_size_t(plan.size), 0, cs)
        else:
            nat.call("ttsk_comm_allreduce_sum", P(sums[slot].ptr), ctypes.c_size_t(plan.size), cs)

    def step_weak():
        slot = counter[0] % inflight
        counter[0] += 1
        if comm_on:
            # the collective path needs the SUM of the rank's sketches only: chains per TT, Psi / Omega contracted over
            # (TT, rank) inside the kernels (no per-TT sketch is written, nothing is summed afterwards); the all-reduce
            # overlaps the products of step s + 1 on the other stream pair
            plan.run_sum(ptr_sets[0], B, sums[slot], stream=2 * slot)
            allreduce_slot(slot, 2 * slot)
            nat.call("ttsk_stream_wait", 2 * slot, cs)      # this slot's streams touch sums again only after its all-reduce
        else:
            plan.run_batch(ptr_sets[0], B, outs[slot], stride, stream=2 * slot)      # stream pair (2 slot, 2 slot + 1)

    def step_strong():
        # the whole job: this rank's passes, then (collective path) one all-reduce of the rank's partial sketch
        if comm_on:
            slot = counter[0] % nslots
            counter[0] += 1
            for i, (b0, cnt) in enumerate(passes):
                plan.run_sum(ptr_sets[i], cnt, sums[slot], accumulate=i > 0, stream=2 * slot)     # accumulating: one after the other
            allreduce_slot(slot, 2 * slot)
            nat.call("ttsk_stream_wait", 2 * slot, cs)      # this slot's streams touch sums[slot] again only after its collective
            return
        for i, (b0, cnt) in enumerate(passes):
            st = 2 * (i % inflight)
            plan.run_batch(ptr_sets[i], cnt, DevArray(outs[0].buf, b0 * stride, (cnt * stride,), (1,)), stride, stream=st)

    step = step_strong if strong else step_weak
    elapsed = job.timed(step, args.steps, args.warmup)
    items_per_step = int(args.items) if strong else B * world
    value = D * items_per_step * args.steps / elapsed

    single_ms, api_ms, cores_per_s_r53 = None, None, None
    if world == 1 and not strong:
        one = (ctypes.c_void_p * plan.d)(*[ptr_sets[0][i] for i in range(plan.d)])
        for _ in range(3):
            plan.run(one, outs[0])
        nat.call("ttsk_sync", -1)
        t1 = time.perf_counter()
        for _ in range(20):
            plan.run(one, outs[0])
        nat.call("ttsk_sync", -1)
        single_ms = 1e3 * (time.perf_counter() - t1) / 20
        # T_total of SURVEY 8d = the reference's own timed region (scripts/experiment_base.py:102-113):
        # stream_sketch() through the Python API incl. DRM sampling, and with to_tt() on top
        import tt_sketch_amd as tsa
        api_ms = {}
        for name, fn in (("stream_sketch", lambda: tsa.stream_sketch(tts[0], left_rank=L_RANK, right_rank=R_RANK)),
                         ("stream_sketch_to_tt", lambda: tsa.stream_sketch(tts[0], left_rank=L_RANK, right_rank=R_RANK).to_tt())):
            best = float("inf")
            for it in range(8):
                nat.call("ttsk_sync", -1)
                t1 = time.perf_counter()
                fn()
                nat.call("ttsk_sync", -1)
                if it >= 2:
                    best = min(best, 1e3 * (time.perf_counter() - t1))
            api_ms[name] = best
        # the reference's other oversampling choice, right rank = left + 3 ("STTA+3", scripts/plot_timings.py:110-124)
        r53 = L_RANK + 3
        right53 = TensorTrainDRM(r53, shape, True, seed=2)
        plan53 = TTSketchPlan(shape, (S_IN,) * (D - 1), left, right53)
        stride53 = plan53.size + (plan53.size & 1)
        out53 = DevArray.empty((B * stride53,))
        for _ in range(3):
            plan53.run_batch(ptr_sets[0], B, out53, stride53, stream=0)
        nat.call("ttsk_sync", -1)
        t1 = time.perf_counter()
        for _ in range(20):
            plan53.run_batch(ptr_sets[0], B, out53, stride53, stream=0)
        nat.call("ttsk_sync", -1)
        cores_per_s_r53 = D * B * 20 / (time.perf_counter() - t1)
        del out53, plan53, right53

    check = None
    if args.check and strong:
        # the sketch of the WHOLE job (sum over every TT of every rank): local sum + one all-reduce, untimed.  The
        # same for every N: what the N = 2 test compares with the N = 1 run.
        if comm_on:
            nat.call("ttsk_sync", -1)
            total = sums[(counter[0] - 1) % nslots]      # the summed sketch of the last step (after a reduce: on rank 0, which reports)
        else:
            total = DevArray.empty((plan.size + (plan.size & 1),))
            if n_mine == 0:
                nat.call("ttsk_memset", P(total.ptr), 0, ctypes.c_size_t(plan.size * 8), 0)
            else:
                for b in range(n_mine):
                    nat.call("ttsk_axpby", P(total.ptr), P(outs[0].ptr + 8 * b * stride), 1.0, 1.0 if b else 0.0,
                             ctypes.c_size_t(plan.size), 0)
        nat.call("ttsk_sync", -1)
        h = total.get()[:plan.size]
        probe = np.random.default_rng(12345).standard_normal(plan.size)
        check = dict(norm=float(np.linalg.norm(h)), probe=float(h @ probe), head=[float(x) for x in h[:4]])

    result = None
    if rank == 0:
        fl = algorithmic_flops(shape, (S_IN,) * (D - 1), (L_RANK,) * (D - 1), (R_RANK,) * (D - 1))
        # ---- roofline leg: per-launch device time of each product class (hipEvents on the launch stream).  The
        # pass runs on ONE stream so that the bracketed times are not inflated by the other chain's kernels.
        os.environ["TTSK_SINGLE_STREAM"] = "1"
        nat.call("ttsk_sync", -1)
        nat.call("ttsk_prof_enable", 1)
        reps = max(5, min(args.steps, 50))
        nb0 = passes[0][1] if passes else 0
        for _ in range(reps if nb0 else 0):
            plan.run_batch(ptr_sets[0], nb0, outs[0], stride, stream=0)
        nat.call("ttsk_sync", -1)
        labels = {0: "right chain GEMM1  T = R^T X^T (two-launch form)",
                  1: "right chain step  R' = sum_k (X_k R) E_k  (fused: both products + slab reduce)",
                  2: "left chain GEMM1  T = L^T X (two-launch form)",
                  3: "left chain step  L' = sum_k (L^T X_k)^T D_k, T stored  (fused: both products + slab reduce)",
                  4: "Psi product  Psi = T R", 5: "small products (Omega, first / last mode)"}
        classes = prof_classes(nat, reps, labels) if nb0 else {}
        os.environ.pop("TTSK_SINGLE_STREAM", None)
        nat.call("ttsk_prof_enable", 0)
        # both roofs per class: algorithmic bytes of one launch (operands read once, result written once, the
        # shared DRM core once per launch) against HBM_TBS, flops against the matrix peak; the lower roof binds
        s_, l_, r_, n_ = S_IN, L_RANK, R_RANK, N_MODE
        mb = {labels[1]: nb0 * (s_ * n_ * s_ + 2 * s_ * r_) + r_ * n_ * r_,
              labels[3]: nb0 * (s_ * n_ * s_ + l_ * n_ * s_ + 2 * s_ * l_) + l_ * n_ * l_,
              labels[4]: nb0 * (l_ * n_ * s_ + l_ * n_ * r_ + s_ * r_)}
        for label, doubles in mb.items():
            if label in classes:
                c = classes[label]
                c["algorithmic_mb_per_launch"] = doubles * 8e-6
                c["algorithmic_tb_per_s"] = doubles * 8 / (c["avg_us"] * 1e-6) * 1e-12
                hbm_roof_tf = c["gflop_per_launch"] * 1e9 / (doubles * 8) * HBM_TBS      # TF/s if bytes moved at HBM_TBS
                c["roof_tflops"] = min(PEAK_F64_MFMA_TF, hbm_roof_tf)
                c["bound"] = "hbm" if hbm_roof_tf < PEAK_F64_MFMA_TF else "mfma"
                c["frac_of_roof"] = c["tflops"] / c["roof_tflops"]
        roofline = None
        if classes:
            dom = max(classes, key=lambda k: classes[k]["share_ms"])
            probe = ctypes.c_double()
            nat.call("ttsk_mfma_f64_peak_probe", ctypes.byref(probe))
            roofline = dict(bound="mfma", kernel=classes[dom]["kernel"], what=dom, achieved=classes[dom]["tflops"],
                            peak=PEAK_F64_MFMA_TF, unit="TFLOP/s", frac=classes[dom]["tflops"] / PEAK_F64_MFMA_TF,
                            traffic=load_traffic(classes[dom]["kernel"]),
                            algorithmic_bytes_per_launch=classes[dom].get("algorithmic_mb_per_launch", 0) * 1e6 or None,
                            avg_launch_us=classes[dom]["avg_us"], probed_mfma_f64_peak=probe.value, classes=classes,
                            pipeline_tflops=fl["total"] * items_per_step * args.steps / elapsed * 1e-12,
                            pipeline_tflops_executed=fl["executed"] * items_per_step * args.steps / elapsed * 1e-12,
                            pipeline_frac_executed=fl["executed"] * items_per_step * args.steps / elapsed * 1e-12 / PEAK_F64_MFMA_TF,
                            pipeline_note="pipeline_tflops counts SURVEY 8d's unfused algorithmic flops (6.017 GF per sketch); "
                                          "_executed leaves out the first product of every interior Psi, which the one-call path "
                                          "shares with the left chain and does not execute")
        cpu, parity = None, None
        if not args.no_cpu and world == 1:      # the CPU leg runs at N = 1 only (the other ranks would sit in the closing barrier)
            h_cores, h_l, h_r = host_cores(tts[0]), [np.asarray(c) for c in left.cores], [np.asarray(c) for c in right.cores]
            cpu, ref = cpu_baseline_tt(shape, h_cores, h_l, h_r, L_RANK, R_RANK)
            if world == 1 and passes:
                plan.run_batch(ptr_sets[0], passes[0][1], outs[0], stride, stream=0)      # the batched result checked below
                nat.call("ttsk_sync", -1)
                import oracle.ttsk_oracle as orc
                ld, rd = orc.TTDrm(h_l, shape, False), orc.TTDrm(h_r, shape, True)
                parity = 0.0
                for b in sorted({0, passes[0][1] - 1}):                    # first and last tensor of the batch
                    rP, rO = (ref if b == 0 else orc.general_sketch("tt", host_cores(tts[b]), ld, rd, "streaming"))
                    want = np.concatenate([a.ravel() for a in rP + rO])
                    got = outs[0].get()[b * stride:b * stride + plan.size]
                    parity = max(parity, float(np.linalg.norm(got - want) / np.linalg.norm(want)))
        result = dict(metric=metric_name(), value=value, unit="TT-cores/s", n_gpus=args.gpus, steps=args.steps,
                      warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
                      scaling="strong" if strong else "weak", vs_baseline=None, dtype="f64", data="synthetic",
                      config=dict(workload="TensorTrain d=6 n=200 TT-rank 100, TensorTrainDRM left rank 50 / right rank 100, "
                                           "streaming sketch (both chains, Omega, Psi), " +
                                           (f"fixed job of {int(args.items)} TTs dealt over the ranks, {B} per batched pass"
                                            if strong else f"{B} TT(s) per GPU per step in one batched pass, {inflight} steps in flight") +
                                           ("; collective path: the rank's TTs are sketched as their SUM (ttsk_tt_sketch_sum: chains per TT, "
                                            "Psi / Omega contracted over (TT, rank) in the kernels) and ONE RCCL " +
                                            ("reduce of one sketch per step leaves the sketch of the whole sum on rank 0" if use_reduce else
                                             "all-reduce of one sketch per step gives every rank the sketch of the whole sum") +
                                            f", {nslots} steps in flight (the collective travels under the next step's products)" if comm_on else ""),
                                  d=D, n=N_MODE, tt_rank=S_IN, left_rank=L_RANK, right_rank=R_RANK,
                                  algorithmic_gflop_per_sketch=fl["total"] * 1e-9, tts_per_step=items_per_step,
                                  steps_in_flight=inflight, single_sketch_latency_ms=single_ms,
                                  t_total_ms_incl_drm_sampling=api_ms, tt_cores_per_s_right_rank_53=cores_per_s_r53,
                                  sketch_bytes=plan.size * 8),
                      roofline=roofline, cpu_baseline=cpu, parity_rel_err_vs_oracle=parity)
        if check is not None:
            result["sketch_check"] = check
    return result



# --------------------------------------------------------------------------- the reference's own published benchmark
# scripts/plot_timings.py:28-36,94-171 -> scripts/results/timings150.csv: shape 100^5, TT-rank 150 (trimmed to
# (100, 150, 150, 100)), sketch rank l = 5 .. 145, right rank 2 l ("x2") or l + 3 ("+3"); the timed region is the
# API call incl. DRM construction (scripts/experiment_base.py:102-113, :132-142, :159-163).  Medians of 20 runs in
# seconds, hardware unstated:
REF150_PUBLISHED_S = {
    "STTAx2": {5: 0.0457, 25: 0.0871, 55: 0.1612, 95: 0.2378, 145: 0.3558},
    "STTA+3": {5: 0.0493, 25: 0.0689, 55: 0.1127, 95: 0.1770, 145: 0.2366},
    "OTTSx2": {5: 0.0495, 25: 0.1567, 55: 0.4247, 95: 0.6941, 145: 1.3204},
    "OTTS+3": {5: 0.0511, 25: 0.1409, 55: 0.2751, 95: 0.5696, 145: 1.0084},
    "HMT": {5: 0.0334, 25: 0.0819, 55: 0.1578, 95: 0.2845, 145: 0.4806},
}
REF150_SHAPE, REF150_TT_RANK = (100,) * 5, (100, 150, 150, 100)


def timed_calls(nat, fn, reps=7, warm=2):
    """best / median wall ms of fn() with the device drained on both sides (results stay in HBM)."""
    times = []
    for it in range(warm + reps):
        nat.call("ttsk_sync", -1)
        t0 = time.perf_counter()
        fn()
        nat.call("ttsk_sync", -1)
        if it >= warm:
            times.append(1e3 * (time.perf_counter() - t0))
    return min(times), float(np.median(times))


def ref150_methods(tsa, tt, l):
    return {"STTAx2": lambda: tsa.stream_sketch(tt, left_rank=l, right_rank=2 * l),
            "STTA+3": lambda: tsa.stream_sketch(tt, left_rank=l, right_rank=l + 3),
            "OTTSx2": lambda: tsa.orthogonal_sketch(tt, left_rank=l, right_rank=2 * l),
            "OTTS+3": lambda: tsa.orthogonal_sketch(tt, left_rank=l, right_rank=l + 3),
            "HMT": lambda: tsa.hmt_sketch(tt, rank=l)}


def chain_classes(nat, fn, reps=5):
    """hipEvent brackets per product class of the TT pipeline while fn() runs on ONE stream."""
    os.environ["TTSK_SINGLE_STREAM"] = "1"
    nat.call("ttsk_sync", -1)
    nat.call("ttsk_prof_enable", 1)
    for _ in range(reps):
        fn()
    nat.call("ttsk_sync", -1)
    labels = {0: "right chain GEMM1 (two-launch form)", 1: "right chain step / GEMM2", 2: "left chain GEMM1 (two-launch form)",
              3: "left chain step / GEMM2", 4: "Psi product", 5: "small products"}
    classes = prof_classes(nat, reps, labels)
    os.environ.pop("TTSK_SINGLE_STREAM", None)
    nat.call("ttsk_prof_enable", 0)
    for c in classes.values():
        c["frac_of_mfma_peak"] = c["tflops"] / PEAK_F64_MFMA_TF
    return classes


def ref150_cpu(tt, l, r, budget_s=12.0):
    """The oracle on the same TT: streaming l / r with pre-built DRMs + T_total incl. sampling, orthogonal, hmt."""
    import __graft_entry__ as ge
    ge.build_oracle()
    from oracle import ttsk_oracle as orc
    shape = REF150_SHAPE
    cores = host_cores(tt)
    rng = np.random.default_rng(0)
    out = {}
    t_end = time.perf_counter() + budget_s
    for name, method in (("STTAx2", "streaming"), ("OTTSx2", "orthogonal"), ("HMT", "hmt")):
        best = float("inf")
        for it in range(3):
          with blas_threads():
            t0 = time.perf_counter()
            trimmed = tuple(min(l, m) for m in (100, 10**4, 10**4, 100))
            ld = None if method == "hmt" else orc.random_tt_drm(shape, trimmed, False, rng)
            rd = orc.random_tt_drm(shape, trimmed if method == "hmt" else r, True, rng)
            orc.general_sketch("tt", cores, ld, rd, method)
            best = min(best, time.perf_counter() - t0)
          if time.perf_counter() > t_end:
                break
        out[name] = best * 1e3
    return out


def ref150_batched(nat, l, r, B, reps=10):
    """Throughput mode at the published shape: B different rank-150 TTs per batched pass (ttsk_tt_sketch_batch), DRMs
    resident -- where the chain kernels can be priced against the matrix peak (one tensor alone is launch-bound)."""
    from tt_sketch_amd import TensorTrainDRM
    from tt_sketch_amd.device import DevArray
    from tt_sketch_amd.tt_fused import TTSketchPlan
    from tt_sketch_amd.utils import process_tt_rank
    shape = REF150_SHAPE
    tts = [device_tt(shape, REF150_TT_RANK, 2000 + b) for b in range(B)]
    lrank = process_tt_rank(l, shape, trim=True)
    left, right = TensorTrainDRM(lrank, shape, False, seed=1), TensorTrainDRM(r, shape, True, seed=2)
    plan = TTSketchPlan(shape, REF150_TT_RANK, left, right)
    stride = plan.size + (plan.size & 1)
    out = DevArray.empty((B * stride,))
    keep, flat = [], []
    for t in tts:
        p1, k1 = plan.core_pointers(t)
        keep.append(k1)
        flat += [p1[i] for i in range(plan.d)]
    ptrs = (ctypes.c_void_p * len(flat))(*flat)
    run = lambda: plan.run_batch(ptrs, B, out, stride, stream=0)
    best, med = timed_calls(nat, run, reps=reps, warm=3)
    classes = chain_classes(nat, run, reps=5)
    fl = algorithmic_flops(shape, REF150_TT_RANK, lrank, (r,) * 4)
    return dict(batch=B, ms_per_pass=med, ms_per_sketch=med / B, tt_cores_per_s=5 * B / (med * 1e-3),
                pipeline_tflops=fl["total"] * B / (med * 1e-3) * 1e-12,
                frac_of_mfma_peak=fl["total"] * B / (med * 1e-3) * 1e-12 / PEAK_F64_MFMA_TF, classes=classes)


def bench_ref150(args, job, ranks=(5, 25, 55, 95, 145), reps=7, cpu=True):
    nat = job.nat
    import tt_sketch_amd as tsa
    shape = REF150_SHAPE
    tt = device_tt(shape, REF150_TT_RANK, 179)
    rows = []
    for l in ranks:
        for name, fn in ref150_methods(tsa, tt, l).items():
            best, med = timed_calls(nat, fn, reps=reps)
            pub = REF150_PUBLISHED_S[name].get(l)
            row = dict(name=name, sketch_rank=l, ms=med, best_ms=best, published_median_s=pub,
                       speedup_vs_published=None if pub is None else pub * 1e3 / med)
            if name.startswith("STTA"):
                r = 2 * l if name == "STTAx2" else l + 3
                from tt_sketch_amd.utils import process_tt_rank
                fl = algorithmic_flops(shape, REF150_TT_RANK, process_tt_rank(l, shape, trim=True), (r,) * 4)
                row.update(algorithmic_gflop=fl["total"] * 1e-9, tflops=fl["total"] / (med * 1e-3) * 1e-12,
                           frac_of_mfma_peak=fl["total"] / (med * 1e-3) * 1e-12 / PEAK_F64_MFMA_TF)
            rows.append(row)
    if job.rank != 0:
        return None
    # the configuration BASELINE.md 1a quotes: l = 55, r = 110
    head_l = 55 if 55 in ranks else ranks[len(ranks) // 2]
    head = next(r for r in rows if r["name"] == "STTAx2" and r["sketch_rank"] == head_l)
    classes = chain_classes(nat, ref150_methods(tsa, tt, head_l)["STTAx2"])
    chain = {k: v for k, v in classes.items() if "chain" in k}
    dom = max(chain or classes, key=lambda k: (chain or classes)[k]["share_ms"])
    dk = classes[dom]
    throughput = ref150_batched(nat, head_l, 2 * head_l, int(args.batch) if args.batch else 16)
    cpu_rec = None
    if cpu and not args.no_cpu and job.world == 1:
        c = ref150_cpu(tt, head_l, 2 * head_l)
        cpu_rec = dict(value=5 / (c["STTAx2"] * 1e-3), unit="TT-cores/s", cores=cpu_cores_used(), kind="port",
                       ms={k: round(v, 1) for k, v in c.items()},
                       sample=f"oracle (same einsum / lstsq / qr calls as the reference) on the same 100^5 rank-150 TT at "
                              f"l={head_l}: DRM sampling + general_sketch, best of 3 each for streaming (r=2l), orthogonal (r=2l), hmt; "
                              f"{CPU_THREADS} BLAS threads")
    return dict(metric=metric_name(), value=5 / (head["ms"] * 1e-3), unit="TT-cores/s", n_gpus=args.gpus, steps=reps, warmup=2,
                ms_per_step=head["ms"], higher_is_better=True, scaling="weak", vs_baseline=head["speedup_vs_published"],
                dtype="f64", data="synthetic",
                config=dict(workload=f"ref150: the reference's published timing sweep (scripts/plot_timings.py:28-36,94-171): TensorTrain "
                                     f"100^5, TT-rank 150 (trimmed {REF150_TT_RANK}), default TensorTrainDRMs, one API call per measurement "
                                     "incl. DRM sampling (the reference's timed region); value / ms_per_step / vs_baseline are the "
                                     f"STTAx2 l={head_l} r={2 * head_l} row; Gaussian cores (the reference's singular-value decay does not "
                                     "change the work)", rows=rows, batched_throughput=throughput),
                roofline=dict(bound="mfma", kernel=dk["kernel"], what=dom + f" at l={head_l} r={2 * head_l}", achieved=dk["tflops"],
                              peak=PEAK_F64_MFMA_TF, unit="TFLOP/s", frac=dk["tflops"] / PEAK_F64_MFMA_TF, traffic=load_traffic(dk["kernel"]),
                              avg_launch_us=dk["avg_us"], classes=classes),
                cpu_baseline=cpu_rec)


# --------------------------------------------------------------------------- C5: TensorSum of 32 rank-20 TTs
def bench_c5(args, job):
    nat = job.nat
    import tt_sketch_amd as tsa
    from tt_sketch_amd.distributed import stream_sketch_sharded
    shape, terms, s, l, r = (128,) * 6, 32, 20, 50, 100
    tts = [device_tt(shape, s, 500 + i) for i in range(terms)]
    S = tsa.TensorSum(tts)
    left = tsa.TensorTrainDRM(l, shape, False, seed=1)
    right = tsa.TensorTrainDRM(r, shape, True, seed=2)
    S.prepare_device()

    root = 0 if getattr(args, "collective", "reduce") == "reduce" else None

    def step():
        if job.comm is not None:
            return stream_sketch_sharded(S, (l,) * 5, (r,) * 5, job.comm, left_drm=left, right_drm=right, root=root)
        return tsa.stream_sketch(S, (l,) * 5, (r,) * 5, left_drm=left, right_drm=right)

    # One rank: the job's throughput as the headline measures it -- sketches of the sum queued on two stream pairs in turn
    # (TTSketchPlan.run_sum = ttsk_tt_sketch_sum, what stream_sketch calls), so the tail of one sketch (joins, small sums) runs beside
    # the chain steps of the next; the single public call is timed separately (`single_call_ms`).
    inflight = 1
    single_call_ms = api_back_to_back_ms = None
    if job.comm is None:
        from tt_sketch_amd.tt_fused import TTSketchPlan
        from tt_sketch_amd.device import DevArray
        plan = TTSketchPlan(shape, tts[0].rank, left, right)
        keep, flat = [], []
        for t in tts:
            p1, k1 = plan.core_pointers(t)
            keep.append(k1)
            flat += [p1[i] for i in range(plan.d)]
        X_all = (ctypes.c_void_p * len(flat))(*flat)
        inflight = int(os.environ.get("TTSK_BENCH_C5_INFLIGHT", "2"))
        outs = [DevArray.empty((plan.size,)) for _ in range(inflight)]
        counter = [0]

        def step_pipelined():
            slot = counter[0] % inflight
            counter[0] += 1
            plan.run_sum(X_all, terms, outs[slot], stream=2 * slot)
        _, single_call_ms = timed_calls(nat, step, reps=15, warm=3)
        nat.call("ttsk_sync", -1)
        t0 = time.perf_counter()
        for _ in range(20):
            step()                                   # the public call back to back on one stream pair (rounds 3-4's number)
        nat.call("ttsk_sync", -1)
        api_back_to_back_ms = 1e3 * (time.perf_counter() - t0) / 20
        elapsed = job.timed(step_pipelined, args.steps, args.warmup)
    else:
        elapsed = job.timed(step, args.steps, args.warmup)
    check = None
    if args.check:
        # the sketch of the whole sum as rank 0 holds it (the same for every N: what the N = 2 test compares with N = 1)
        stt = step()
        h = np.concatenate([np.asarray(a).ravel() for a in stt.Psi_cores + stt.Omega_mats])
        probe = np.random.default_rng(12345).standard_normal(h.size)
        check = dict(norm=float(np.linalg.norm(h)), probe=float(h @ probe), head=[float(x) for x in h[:4]], size=int(h.size))
        job.barrier()
    if job.rank != 0:
        return None
    fl = algorithmic_flops(shape, (s,) * 5, (l,) * 5, (r,) * 5)
    gf = fl["total"] * terms
    t_step = elapsed / args.steps
    cpu = None
    if not args.no_cpu and job.world == 1:
        cpu, _ = cpu_baseline_tt(shape, host_cores(tts[0]), [np.asarray(c) for c in left.cores],
                                 [np.asarray(c) for c in right.cores], l, r, budget_s=6.0)
        # the reference sketches a sum term by term (sketch_dispatch.py:85-139): 32 x the per-term time
        cpu["value"] = 6 * terms / (terms * cpu["t_sketch_ms"] * 1e-3)
        cpu["sample"] = "ONE of the 32 terms timed, the sum priced as 32 such sketches (sketch_dispatch.py:85-139): " + cpu["sample"]
    return dict(metric=metric_name(), value=6 * terms / t_step, unit="TT-cores/s", n_gpus=args.gpus, steps=args.steps,
                warmup=args.warmup, ms_per_step=1e3 * t_step, higher_is_better=True, scaling="strong", vs_baseline=None,
                dtype="f64", data="synthetic",
                config=dict(workload="C5: stream_sketch of a TensorSum of 32 rank-20 TTs, d=6 n=128, shared TensorTrainDRMs l=50 r=100; "
                                     "terms dealt over the ranks, one all-reduce (stream_sketch_sharded)" , terms=terms,
                            steps_in_flight=inflight, single_call_ms=single_call_ms, api_back_to_back_ms=api_back_to_back_ms),
                roofline=dict(bound="mfma", kernel="ttsk_tt_sketch_sum of the 32 terms: chain_sum_kernel (stacked-terms chain steps: 16-row tiles that span terms), "
                                                   "Psi of the sum as K chunks on stream_small_kernel, the Omega of all modes in one launch",
                              achieved=gf / t_step * 1e-12, peak=PEAK_F64_MFMA_TF, unit="TFLOP/s",
                              frac=gf / t_step * 1e-12 / PEAK_F64_MFMA_TF, traffic=load_traffic("c5_sketch"),
                              algorithmic_bytes=(terms * 13.9e6 + 60.0e6),
                              what="algorithmic flops of the 32 term sketches (SURVEY 8d: 0.443 GF each) / time per sketch with two sketches in "
                                   "flight on two stream pairs (one rank; as the headline's steps_in_flight); config.single_call_ms: one public "
                                   "stream_sketch call incl. Python, drained on both sides; api_back_to_back_ms: such calls one after the other"),
                cpu_baseline=cpu, **({"sketch_check": check} if check is not None else {}))


# --------------------------------------------------------------------------- C2: dense d=5 n=64
def bench_c2(args, job, gaussian=False):
    """C2, TensorTrainDRM variant (default) or DenseGaussianDRM variant (device-sampled matrices, SURVEY 8d names both)."""
    nat = job.nat
    import tt_sketch_amd as tsa
    from tt_sketch_amd.utils import random_normal_dev
    shape, l, r = (64,) * 5, 20, 40
    X = random_normal_dev(shape, seed=2)
    T = tsa.DenseTensor(X)
    cls = tsa.DenseGaussianDRM if gaussian else tsa.TensorTrainDRM
    left = cls(l, shape, False, seed=1)
    right = cls(r, shape, True, seed=2)

    def step():
        tsa.general_sketch(T, left, right, tsa.SketchMethod.streaming)
    elapsed = job.timed(step, args.steps, args.warmup)
    if job.rank != 0:
        return None
    t_step = elapsed / args.steps
    x_bytes, mats_bytes, unfused, gflop_ref = 8.59e9, 8.18e9, 77.3e9, 484.0
    cpu = None
    if not args.no_cpu and job.world == 1:
        # the oracle's dense path at full size needs c_einsum over 8.6 GB (minutes): the largest size that
        # finishes in seconds, same ranks -- n = 32 (268 MB), 1/32 of the entries
        import __graft_entry__ as ge
        ge.build_oracle()
        from oracle import ttsk_oracle as orc
        rng = np.random.default_rng(2)
        shp = (32,) * 5
        Xs = rng.standard_normal(shp)
        if gaussian:
            ld = orc.DenseDrm([rng.standard_normal((l, 32 ** (mu + 1))) for mu in range(4)], shp, False)
            rd = orc.DenseDrm([rng.standard_normal((r, 32 ** (mu + 1))) for mu in range(4)], shp, True)
        else:
            ld, rd = orc.random_tt_drm(shp, l, False, rng), orc.random_tt_drm(shp, r, True, rng)
        with blas_threads():
            t0 = time.perf_counter()
            orc.general_sketch("dense", Xs, ld, rd, "streaming")
            t_cpu = time.perf_counter() - t0
        cpu = dict(value=5 / t_cpu, unit="TT-cores/s", cores=cpu_cores_used(), kind="port", t_sketch_ms=t_cpu * 1e3,
                   gb_per_s=Xs.nbytes / t_cpu * 1e-9,
                   sample=f"oracle general_sketch of a dense d=5 n=32 tensor (268 MB = 1/32 of C2), {'DenseGaussianDRM' if gaussian else 'TensorTrainDRM'} l=20 r=40, "
                          f"DRMs pre-built, one run (~1 s); {CPU_THREADS} BLAS threads")
        del Xs
    # just outside the round-3 cover of the one-pass kernel (left rank <= 20, even right rank <= 40): the same tensor, wider DRMs
    cover = None
    if not gaussian and getattr(args, "cover", False):
        cover = [dict(drm="TensorTrainDRM l=20 r=40 (in cover)", ms=1e3 * t_step)]
        for lw, rw in ((21, 42), (20, 41), (32, 64)):
            lw_drm, rw_drm = tsa.TensorTrainDRM(lw, shape, False, seed=1), tsa.TensorTrainDRM(rw, shape, True, seed=2)
            _, med = timed_calls(nat, lambda: tsa.general_sketch(T, lw_drm, rw_drm, tsa.SketchMethod.streaming), reps=7, warm=3)
            cover.append(dict(drm=f"TensorTrainDRM l={lw} r={rw}", ms=med, vs_in_cover=med / (1e3 * t_step)))
    base = dict(metric=metric_name(), value=5 / t_step, unit="TT-cores/s", n_gpus=args.gpus, steps=args.steps,
                warmup=args.warmup, ms_per_step=1e3 * t_step, higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype="f64", data="synthetic", cpu_baseline=cpu)
    if gaussian:
        # materialised Gaussian matrices: the binding roof is HBM -- SURVEY 8d's one-pass bytes are the tensor once plus
        # every DRM matrix once
        alg = x_bytes + mats_bytes
        traffic = load_traffic("c2_gaussian_sketch")
        base.update(config=dict(workload="C2 (DenseGaussianDRM): general_sketch of a dense fp64 tensor d=5 n=64 (8.59 GB resident), device-sampled DenseGaussianDRM l=20 r=40 "
                                         "(8.18 GB of matrices)"),
                    roofline=dict(bound="hbm", kernel="dense_gauss_pass_kernel (every left product A_mu X^{<mu+1>} and Psi_0 from ONE read of X; the right-hand products read the small left products)",
                                  achieved=alg / t_step * 1e-9, peak=HBM_TBS * 1e3, unit="GB/s", frac=alg / t_step / (HBM_TBS * 1e12),
                                  traffic=traffic, traffic_over_algorithmic=None if traffic is None else traffic / alg,
                                  algorithmic_bytes=alg,
                                  what="SURVEY 8d one-pass bytes (8.59 GB tensor + 8.18 GB of DRM matrices) / wall time of one sketch",
                                  mfma_frac_of_algorithmic_flops=gflop_ref / t_step * 1e-3 / PEAK_F64_MFMA_TF,
                                  algorithmic_tflops=gflop_ref / t_step * 1e-3))
        return base
    # TensorTrainDRM recipes: no DRM matrix beyond 84 MB exists, so the bytes that must move are the tensor once (+ the sketch
    # itself, < 1 MB); the kernels execute 129 GF (pass over X) + 41 GF (pass over the first left product) of the 484 GF the
    # reference's formulation counts, and the matrix pipe -- not HBM -- is the roof that binds
    alg = x_bytes
    executed_gf = 170.0
    traffic = load_traffic("c2_sketch")
    hbm_frac = alg / t_step / (HBM_TBS * 1e12)
    mfma_frac = executed_gf / t_step * 1e-3 / PEAK_F64_MFMA_TF
    base.update(config=dict(workload="C2: general_sketch of a dense fp64 tensor d=5 n=64 (8.59 GB resident), TensorTrainDRM l=20 r=40"),
                roofline=dict(bound="mfma", kernel="dense_pass_kernel (Z_0 and Psi_0 from one read of X, Z_1 and Psi_1 from one read of Z_0; 90 % of the sketch)",
                              achieved=executed_gf / t_step * 1e-3, peak=PEAK_F64_MFMA_TF, unit="TFLOP/s", frac=mfma_frac,
                              traffic=traffic, traffic_over_algorithmic=None if traffic is None else traffic / alg,
                              algorithmic_bytes=alg, hbm_gbs=alg / t_step * 1e-9, hbm_frac=hbm_frac,
                              executed_gflop=executed_gf, reference_count_gflop=gflop_ref,
                              mfma_frac_of_reference_count=gflop_ref / t_step * 1e-3 / PEAK_F64_MFMA_TF,
                              what="executed flops (129 GF over X + 41 GF over the first left product; the recipes need 170 of the 484 GF of the "
                                   "reference's formulation) / wall time of one sketch against the fp64 matrix peak; hbm_*: the 8.59 GB that must "
                                   "move (the tensor once; no DRM matrix beyond 84 MB is formed) / the same time; counter traffic also holds the "
                                   "first left product written and read once (2 x 2.7 GB) and the partial sums",
                              unfused_77GB_rate_gbs=unfused / t_step * 1e-9, cover=cover))
    return base


# --------------------------------------------------------------------------- C4: sparse 1e7 nnz
def bench_c4(args, job):
    import tt_sketch_amd as tsa
    shape, nnz, l, r = (200, 150, 100, 120, 300), 10_000_000, 10, 15
    rng = np.random.default_rng(4)
    idx = np.stack([rng.integers(0, n, nnz) for n in shape]).astype(np.int64)
    T = tsa.SparseTensor(shape, idx, rng.standard_normal(nnz))
    left = tsa.SparseGaussianDRM(l, shape, False, seed=3)
    right = tsa.SparseGaussianDRM(r, shape, True, seed=4)
    nat = job.nat
    # first-call cost of a NEW sparse tensor (the reference's actual use): H2D of indices + values, the per-mode
    # orderings (ttsk_sparse_sort_mode), and the first sketch
    nat.call("ttsk_sync", -1)
    t0 = time.perf_counter()
    T.prepare_device()
    nat.call("ttsk_sync", -1)
    t_h2d = time.perf_counter() - t0
    t0 = time.perf_counter()
    tsa.general_sketch(T, left, right, tsa.SketchMethod.streaming)
    nat.call("ttsk_sync", -1)
    t_first = time.perf_counter() - t0

    def step():
        tsa.general_sketch(T, left, right, tsa.SketchMethod.streaming)
    elapsed = job.timed(step, args.steps, args.warmup)
    if job.rank != 0:
        return None
    t_step = elapsed / args.steps
    nbytes, samples = 8.0 * nnz * 6, (l + r) * 4 * nnz
    # per kernel class: hipEvent brackets (ttsk_prof_*) around the sampler passes and the segmented sums
    nat.call("ttsk_prof_enable", 1)
    reps = 3
    for _ in range(reps):
        step()
    nat.call("ttsk_sync", -1)
    classes = prof_classes(nat, reps, {6: "hash-Gaussian sampling into panels / tables (fast_lazy_gaussian.pyx:52-105,183-202)",
                                       7: "one pass per mode: stream + DRM rows + segmented sums (sparse_sketch.py:8-69)"})
    nat.call("ttsk_prof_enable", 0)
    from tt_sketch_amd import sparse_fused
    plan = dict(sparse_fused.last_plan)
    # VALU roof of a Gaussian sample, MEASURED on this device (ttsk_ndtri_rate_probe: hash + ndtri with the kernels' central / tail
    # split on resident operands, nothing else; [1] = every lane through both branches).  (Rounds 2-3 priced against an
    # instruction-count estimate of 393 G samples / s.)
    probe = (ctypes.c_double * 2)()
    nat.call("ttsk_ndtri_rate_probe", probe)
    VALU_GSAMPLES = float(probe[0])
    for label, c in classes.items():
        work = c.pop("gflop_per_launch") * 1e9          # the class's own work unit per launch
        c.pop("tflops", None)
        if label.startswith("hash"):
            c.update(bound="valu", samples_per_launch=work, gsamples_per_s=work / (c["avg_us"] * 1e-6) * 1e-9,
                     peak_gsamples_per_s=VALU_GSAMPLES, frac=work / (c["avg_us"] * 1e-6) * 1e-9 / VALU_GSAMPLES)
        else:
            per_pass = plan.get("sampled_columns_per_nonzero", 0) * nnz / max(plan.get("passes", 1), 1)
            c.update(bound="valu" if per_pass else "hbm", stream_mb_per_launch=work * 1e-6, stream_gb_s=work / (c["avg_us"] * 1e-6) * 1e-9,
                     hbm_frac=work / (c["avg_us"] * 1e-6) / (HBM_TBS * 1e12),
                     gsamples_per_s=per_pass / (c["avg_us"] * 1e-6) * 1e-9, peak_gsamples_per_s=VALU_GSAMPLES,
                     frac=(per_pass / (c["avg_us"] * 1e-6) * 1e-9 / VALU_GSAMPLES) if per_pass else work / (c["avg_us"] * 1e-6) / (HBM_TBS * 1e12),
                     traffic=load_traffic(c["kernel"]), plan=plan)
    # just outside the round-3 cover (<= 16 columns per factor, SparseGaussianDRM only): the same tensor with wider and with
    # sign DRMs, a few sketches each; "ms_per_column" = time / DRM columns made per nonzero, the in-cover row first
    cover = []
    G, S = tsa.SparseGaussianDRM, tsa.SparseSignDRM
    for name, mk in [] if not getattr(args, "cover", False) else [("gaussian l=10 r=15 (in cover)", lambda: (G(l, shape, False, seed=3), G(r, shape, True, seed=4))),
                     ("gaussian l=17 r=17", lambda: (G(17, shape, False, seed=3), G(17, shape, True, seed=4))),
                     ("gaussian l=24 r=24", lambda: (G(24, shape, False, seed=3), G(24, shape, True, seed=4))),
                     ("sign l=10 r=15", lambda: (S(l, shape, False, seed=3), S(r, shape, True, seed=4))),
                     ("sign l=24 r=24", lambda: (S(24, shape, False, seed=3), S(24, shape, True, seed=4)))]:
        ld, rd = mk()
        _, med = timed_calls(nat, lambda: tsa.general_sketch(T, ld, rd, tsa.SketchMethod.streaming), reps=3)
        cols = sparse_fused.last_plan.get("sampled_columns_per_nonzero", 0)
        cover.append(dict(drm=name, ms=med, columns_per_nonzero=cols, ms_per_column=med / max(cols, 1)))
    for c in cover:
        c["per_column_vs_in_cover"] = c["ms_per_column"] / cover[0]["ms_per_column"]
    cover = cover or None
    cpu = None
    if not args.no_cpu and job.world == 1:
        # the reference's Psi is O(n_mu nnz) boolean masks (sparse_sketch.py:18,60): nnz = 2e5 takes seconds, 1e7 minutes
        import __graft_entry__ as ge
        ge.build_oracle()
        from oracle import ttsk_oracle as orc
        ns = 200_000
        t0 = time.perf_counter()
        orc.general_sketch("sparse", (shape, idx[:, :ns].copy(), T.entries[:ns].copy()),
                           orc.HashGaussDrm(3, shape, False, (0,) * 4, (l,) * 4), orc.HashGaussDrm(4, shape, True, (0,) * 4, (r,) * 4),
                           "streaming")
        t_cpu = time.perf_counter() - t0
        cpu = dict(value=5 / t_cpu, unit="TT-cores/s", cores=1, kind="port", t_sketch_ms=t_cpu * 1e3, nnz=ns,
                   nnz_per_s=ns / t_cpu,
                   sample=f"oracle general_sketch (C hash sampler + NumPy masks) on the first {ns} nonzeros of the same tensor (1/50 of C4; "
                          "the reference's Psi is O(n_mu nnz), so full size extrapolates to minutes); single thread as the reference")
    return dict(metric=metric_name(), value=5 / t_step, unit="TT-cores/s", n_gpus=args.gpus, steps=args.steps,
                warmup=args.warmup, ms_per_step=1e3 * t_step, higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype="f64", data="synthetic",
                config=dict(workload="C4: general_sketch of a COO tensor d=5 shape (200,150,100,120,300) nnz=1e7 (resident), "
                                     "SparseGaussianDRM l=10 r=15"),
                roofline=dict(bound="hbm", kernel="sg_pass_kernel (5 x: one per mode)",
                              achieved=nbytes / t_step * 1e-9, peak=HBM_TBS * 1e3, unit="GB/s",
                              frac=nbytes / t_step / (HBM_TBS * 1e12), traffic=load_traffic("c4_sketch"),
                              what="SURVEY 8d bytes 8*nnz*(d+1) = 480 MB / wall time of one sketch (the per-class entry prices the passes "
                                   "against the fp64 VALU, which binds: every DRM row of a deep mode is an ndtri evaluation)",
                              gaussian_samples_per_s=samples / t_step, classes=classes,
                              measured_valu_gsamples=dict(split=float(probe[0]), divergent=float(probe[1])), cover=cover,
                              first_call_ms=dict(h2d_and_upload=t_h2d * 1e3, first_sketch_incl_mode_sorts=t_first * 1e3,
                                                 steady_state=t_step * 1e3)),
                cpu_baseline=cpu)


# --------------------------------------------------------------------------- the solves at C3
def bench_c3_solves(args, job, reps=7):
    """orthogonal_sketch / hmt_sketch / stream_sketch().to_tt() at the C3 shape through the public API (incl. DRM sampling, the
    reference's timed region), with the oracle (scipy lstsq / qr as the reference) on the host beside them."""
    nat = job.nat
    import tt_sketch_amd as tsa
    shape = (N_MODE,) * D
    tt = device_tt(shape, S_IN, 77)
    calls = {"orthogonal_sketch": lambda: tsa.orthogonal_sketch(tt, left_rank=L_RANK, right_rank=R_RANK),
             "hmt_sketch": lambda: tsa.hmt_sketch(tt, rank=L_RANK),
             "stream_sketch_to_tt": lambda: tsa.stream_sketch(tt, left_rank=L_RANK, right_rank=R_RANK).to_tt()}
    fl = algorithmic_flops(shape, (S_IN,) * (D - 1), (L_RANK,) * (D - 1), (R_RANK,) * (D - 1))
    m = L_RANK * N_MODE
    # beyond the sketch's products: pinv-apply 2 m r l per mode and (orthogonal / hmt) thin QR 4 m l^2 - 4/3 l^3 per mode
    extra = {"orthogonal_sketch": (D - 1) * (2 * m * R_RANK * L_RANK + 4 * m * L_RANK**2),
             "hmt_sketch": (D - 1) * 4 * m * L_RANK**2 - fl["left"] - fl["omega"], "stream_sketch_to_tt": (D - 1) * 2 * m * R_RANK * L_RANK}
    out = {}
    for name, fn in calls.items():
        best, med = timed_calls(nat, fn, reps=reps)
        gf = (fl["total"] + extra[name]) * 1e-9
        out[name] = dict(ms=med, best_ms=best, algorithmic_gflop=gf,
                         roofline=dict(bound="mfma", achieved=gf / med, peak=PEAK_F64_MFMA_TF, unit="TFLOP/s", frac=gf / med / PEAK_F64_MFMA_TF,
                                       traffic=None, what="sketch products + pinv-apply + thin QR flops / wall time of one API call incl. DRM "
                                                          "sampling: a chain of ~80 dependent launches on 10^4-row operands (ttsk_tt_orth_sketch), latency-bound"))
    # same-signature tensors against ONE pair of DRMs: concurrent chains on the library's stream pairs
    tts = [device_tt(shape, S_IN, 78 + b) for b in range(8)]
    ldrm = tsa.TensorTrainDRM(L_RANK, shape, False, seed=1)
    rdrm = tsa.TensorTrainDRM(R_RANK, shape, True, seed=2)
    hdrm = tsa.TensorTrainDRM(L_RANK, shape, True, seed=3)
    lr, rr = (L_RANK,) * (D - 1), (R_RANK,) * (D - 1)
    for name, fn, single in (("orthogonal_sketch_batch8", lambda: tsa.orthogonal_sketch_batch(tts, lr, rr, left_drm=ldrm, right_drm=rdrm),
                              lambda: tsa.orthogonal_sketch(tts[0], lr, rr, left_drm=ldrm, right_drm=rdrm)),
                             ("hmt_sketch_batch8", lambda: tsa.hmt_sketch_batch(tts, lr, drm=hdrm), lambda: tsa.hmt_sketch(tts[0], lr, drm=hdrm))):
        best, med = timed_calls(nat, fn, reps=reps)
        _, one = timed_calls(nat, single, reps=reps)
        out[name] = dict(ms=med, best_ms=best, ms_per_tensor=med / len(tts), one_call_with_the_same_drms_ms=one, tensors=len(tts),
                         what="8 TTs of the C3 signature, DRMs given (no sampling in the timed region): ttsk_tt_orth_sketch_batch")
    del tts
    if not args.no_cpu and job.world == 1 and job.rank == 0:
        import __graft_entry__ as ge
        ge.build_oracle()
        from oracle import ttsk_oracle as orc
        cores = host_cores(tt)
        rng = np.random.default_rng(0)
        ld, rd = orc.random_tt_drm(shape, L_RANK, False, rng), orc.random_tt_drm(shape, R_RANK, True, rng)
        rdh = orc.random_tt_drm(shape, L_RANK, True, rng)
        with blas_threads():
            t0 = time.perf_counter(); orc.general_sketch("tt", cores, ld, rd, "orthogonal"); t_o = time.perf_counter() - t0
            t0 = time.perf_counter(); orc.general_sketch("tt", cores, None, rdh, "hmt"); t_h = time.perf_counter() - t0
            t0 = time.perf_counter(); P, O = orc.general_sketch("tt", cores, ld, rd, "streaming"); orc.assemble(P, O); t_t = time.perf_counter() - t0
        for name, t in (("orthogonal_sketch", t_o), ("hmt_sketch", t_h), ("stream_sketch_to_tt", t_t)):
            out[name]["cpu_baseline"] = dict(value=D / t, unit="TT-cores/s", cores=cpu_cores_used(), kind="port", ms=t * 1e3,
                                             sample="oracle (einsum + scipy.linalg.lstsq / qr as sketch_dispatch.py:160-193, sketch.py:400-443) on the "
                                                    f"same C3 tensor, DRMs pre-built, ONE run; {CPU_THREADS} BLAS threads")
            out[name]["value"] = D / (out[name]["ms"] * 1e-3)
            out[name]["unit"] = "TT-cores/s"
    return out


def compact(line):
    """sub-record of a full bench line: what the judge's table needs, nothing else"""
    if line is None:
        return None
    roof = {k: line["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "algorithmic_bytes",
                                                  "hbm_gbs", "hbm_frac", "executed_gflop", "mfma_frac_of_reference_count", "mfma_frac_of_algorithmic_flops",
                                                  "classes", "first_call_ms", "measured_valu_gsamples", "cover")
            if line.get("roofline") and k in line["roofline"]}
    cpu = line.get("cpu_baseline")
    if cpu:
        cpu = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "sample", "t_sketch_ms", "ms") if k in cpu}
    out = dict(workload=line["config"]["workload"], ms=line["ms_per_step"], value=line["value"], unit=line["unit"], roofline=roof,
               cpu_baseline=cpu)
    for k in ("steps_in_flight", "single_call_ms", "api_back_to_back_ms"):
        if k in line["config"]:
            out[k] = line["config"][k]
    return out


def run_extras(args, job):
    """The other BASELINE configurations and the solves, measured in the SAME default run the driver records:
    c2 / c4 / c5, orthogonal / hmt / to_tt at C3, the reference's published rank-150 rows."""
    import copy
    extra = {}
    sub = copy.copy(args)
    sub.steps, sub.warmup = 5, 2
    sub.cover = True
    for name, fn in (("c5", bench_c5), ("c4", bench_c4), ("c2", bench_c2), ("c2_gaussian", lambda a, j: bench_c2(a, j, gaussian=True))):
        t0 = time.perf_counter()
        # (c2: a sketch is 4 ms and its first calls grow the library's scratch arenas by 1.3 GB -- a few more of both)
        # (c5: half a millisecond per sketch, two in flight -- enough of them for the pipeline to fill)
        sub.steps, sub.warmup = (10, 3) if name.startswith("c2") else (30, 4) if name == "c5" else (5, 2)
        try:
            extra[name] = compact(fn(sub, job))
            extra[name]["bench_wall_s"] = time.perf_counter() - t0
        except Exception as e:          # a failing side leg must not take the headline line with it
            extra[name] = dict(error=f"{type(e).__name__}: {e}")
        from tt_sketch_amd.device import release_cached
        release_cached()
    sub.steps, sub.warmup = 5, 2
    try:
        extra["c3_solves"] = bench_c3_solves(sub, job)
    except Exception as e:
        extra["c3_solves"] = dict(error=f"{type(e).__name__}: {e}")
    try:
        sub.batch = 32
        line = bench_ref150(sub, job, ranks=(55,), reps=5)
        extra["ref150"] = dict(rows=[{k: r[k] for k in ("name", "sketch_rank", "ms", "published_median_s", "speedup_vs_published")}
                                     for r in line["config"]["rows"]],
                               batched_throughput={k: v for k, v in line["config"]["batched_throughput"].items() if k != "classes"},
                               dominant_chain_kernel_batched=max(
                                   ({"what": k, **{q: v[q] for q in ("kernel", "avg_us", "tflops", "frac_of_mfma_peak")}}
                                    for k, v in line["config"]["batched_throughput"]["classes"].items() if "chain" in k),
                                   key=lambda c: c["tflops"], default=None),
                               cpu_baseline=line["cpu_baseline"])
    except Exception as e:
        extra["ref150"] = dict(error=f"{type(e).__name__}: {e}")
    return extra


def main():
    # dmabuf IPC is the only mode the host driver supports (RCCL between processes); must precede HIP start-up
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=("c3", "c2", "c2g", "c4", "c5", "ref150"), default="c3")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="default: weak on one GPU; with --gpus N > 1 the headline is the STRONG job (north_star's quantity) and the "
                         "weak number rides along as a sub-record")
    ap.add_argument("--collective", choices=("reduce", "allreduce"), default="reduce",
                    help="collective path: one reduce to rank 0 (default) or an all-reduce")
    ap.add_argument("--items", type=int, default=128, help="--scaling strong: TTs in the fixed job")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--check", action="store_true",
                    help="--scaling strong: also print norm / probe of the sketch of the whole job (identical for every N)")
    ap.add_argument("--cover", action="store_true", help="--config c4: also time the shapes just outside the round-3 cover (wider / sign DRMs); "
                                                         "always on in the default run's c4 sub-record")
    ap.add_argument("--no-extra", action="store_true", help="default config only: skip the c2 / c4 / c5 / solves / ref150 sub-records")
    ap.add_argument("--batch", type=int, default=32, help="TTs per batched pass (ttsk_tt_sketch_batch; 32 = 8 workgroups x 25 slices per tensor in the fused chain step)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="independent passes in flight (issued on alternating stream pairs); 1 = strictly one after the other")
    args = ap.parse_args()
    if args.config != "c3" and args.steps == 200:
        args.steps, args.warmup = 20, 3
    job = Job(args)
    try:
        both = args.scaling is None and args.config == "c3" and job.world > 1
        if args.scaling is None:
            args.scaling = "strong" if both else "weak"
        result = {"c3": bench_c3, "c2": bench_c2, "c2g": lambda a, j: bench_c2(a, j, gaussian=True), "c4": bench_c4, "c5": bench_c5,
                  "ref150": bench_ref150}[args.config](args, job)
        if (result is not None and args.config == "c3" and not args.no_extra and job.world == 1
                and args.scaling == "weak" and not os.environ.get("TTSK_BENCH_FORCE_COMM")):
            result["extra"] = run_extras(args, job)
        if both:
            # the same launch also measures the weak job (per-GPU work fixed): a sub-record of the one line
            import copy
            sub = copy.copy(args)
            sub.scaling, sub.no_cpu, sub.check = "weak", True, False
            sub.steps, sub.warmup = min(args.steps, 60), min(args.warmup, 10)
            weak = bench_c3(sub, job)
            if result is not None and weak is not None:
                result["weak"] = {k: weak[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "scaling")}
                result["weak"]["workload"] = weak["config"]["workload"]
        if result is not None:
            print(json.dumps(result))
    finally:
        job.close()            # the communicator and the rendezvous files go away on every path, failed or not
    return 0


if __name__ == "__main__":
    sys.exit(main())
