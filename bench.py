"""Benchmark: TT-cores sketched / second (fp64) of stream_sketch on the north-star workload.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] / SURVEY.md 8d "C3"): TensorTrain d=6, n=200, TT-rank 100,
TensorTrainDRM left rank 50 (r_out), right rank 100, all fp64, synthetic Gaussian cores.  A step
sketches one such TT per GPU with inputs and DRMs resident in HBM (ttsk_tt_sketch: both chains,
Omega, Psi).  With N > 1 ranks every rank sketches its own summand and ONE RCCL all-reduce sums
the packed partial sketches (the sketch of the N-term TensorSum) -- weak scaling, value = cores
sketched by all ranks per second.

The JSON line also carries
  roofline      the dominant GEMM class of the pipeline: algorithmic flops / hipEvent time per
                launch against the fp64 MFMA peak (78.6 TF/s data sheet; probed ceiling reported)
  cpu_baseline  the CPU oracle (NumPy restatement of the reference, same einsum calls) on the
                same inputs, on this box's host cores
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
PEAK_F64_MFMA_TF = 78.6   # AMD MI355X data sheet (fp64 matrix); not listed in MI355X_MICROARCH.md


def algorithmic_flops(shape, s, l, r):
    """cheapest pairwise order of every 3-operand einsum, unfused."""
    d = len(shape)
    S = (1,) + tuple(s) + (1,)
    Lr = (1,) + tuple(l)
    Rr = tuple(r) + (1,)

    def chain(rk):
        tot = 2 * shape[0] * S[1] * rk[1]
        for mu in range(1, d - 1):
            a, n, b, p, q = S[mu], shape[mu], S[mu + 1], rk[mu], rk[mu + 1]
            tot += min(2 * a * p * n * b + 2 * p * n * b * q, 2 * a * p * n * q + 2 * a * n * b * q)
        return tot
    left = chain(Lr)
    # right chain walks the transposed tensor
    St, shp_t, Rt = S[::-1], shape[::-1], (1,) + tuple(r[::-1])
    right = 2 * shp_t[0] * St[1] * Rt[1]
    for mu in range(1, d - 1):
        a, n, b, p, q = St[mu], shp_t[mu], St[mu + 1], Rt[mu], Rt[mu + 1]
        right += min(2 * a * p * n * b + 2 * p * n * b * q, 2 * a * p * n * q + 2 * a * n * b * q)
    omega = sum(2 * l[mu] * S[mu + 1] * r[mu] for mu in range(d - 1))
    psi = 2 * shape[0] * S[1] * Rr[0] + 2 * Lr[d - 1] * S[d - 1] * shape[d - 1]
    for mu in range(1, d - 1):
        a, n, b, p, q = S[mu], shape[mu], S[mu + 1], Lr[mu], Rr[mu]
        psi += min(2 * p * a * n * b + 2 * p * n * b * q, 2 * a * n * b * q + 2 * p * a * n * q)
    return dict(left=left, right=right, omega=omega, psi=psi, total=left + right + omega + psi)


def load_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_traffic.json, written by profiles/collect_traffic.py: FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for 16-byte streaming reads on gfx950, plus WRITE_SIZE)."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    entry = table.get(kernel_name)
    return entry["bytes_per_launch"] if entry else None


def make_inputs(seed):
    rng = np.random.default_rng(seed)
    shape = (N_MODE,) * D
    S = (1,) + (S_IN,) * (D - 1) + (1,)
    cores = [rng.standard_normal((S[k], shape[k], S[k + 1])) / np.sqrt(S[k] * shape[k]) for k in range(D)]

    def drm_cores(rank):
        rk = (1,) + (rank,) * (D - 1)
        return [rng.standard_normal((rk[k], N_MODE, rk[k + 1])) / np.sqrt(rk[k]) for k in range(D - 1)]
    return shape, cores, drm_cores(L_RANK), drm_cores(R_RANK)


def cpu_baseline(shape, cores, lcores, rcores, budget_s=12.0):
    """Oracle (port of the reference path, same einsum strings/optimize flags) on host cores."""
    import __graft_entry__ as ge
    ge.build_oracle()
    from oracle import ttsk_oracle as orc
    ld = orc.TTDrm(lcores, shape, False)
    rd = orc.TTDrm(rcores, shape, True)
    orc.general_sketch("tt", cores, ld, rd, "streaming")       # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 40):
        t0 = time.perf_counter()
        res = orc.general_sketch("tt", cores, ld, rd, "streaming")
        times.append(time.perf_counter() - t0)
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    best = min(times)
    return dict(value=D / best, unit="TT-cores/s", cores=int(threads), kind="port",
                sample=f"{len(times)} sketches of the same d={D} n={N_MODE} s={S_IN} l={L_RANK} r={R_RANK} TT, "
                       f"DRMs pre-built; best {best * 1e3:.1f} ms, median {np.median(times) * 1e3:.1f} ms",
                ), res


def main():
    # dmabuf IPC is the only mode the host driver supports (RCCL between processes); must precede HIP start-up
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--batch", type=int, default=16, help="TTs per step, sketched in one batched pass (ttsk_tt_sketch_batch)")
    ap.add_argument("--graph", type=int, default=0, help="replay the step from a hipGraph (1) or launch eagerly (0)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="independent sketches in flight (items of the tensor stream are issued on alternating "
                         "stream pairs); 1 = strictly one after the other")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world

    from tt_sketch_amd import _native as nat
    from tt_sketch_amd import TensorTrain, TensorTrainDRM
    from tt_sketch_amd.tt_fused import TTSketchPlan
    nat.call("ttsk_init", int(os.environ.get("TTSK_BENCH_DEVICE", local_rank)))   # override: rehearsals on a one-GPU box

    dist = None
    comm_on = world > 1 or bool(os.environ.get("TTSK_BENCH_FORCE_COMM"))   # rehearsal: the collective path with one rank
    if world == 1 and comm_on:
        uid = (ctypes.c_char * 128)()
        nat.call("ttsk_comm_unique_id", uid)
        nat.call("ttsk_comm_init", uid, 0, 1)
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # host-side rendezvous only
        uid = (ctypes.c_char * 128)()
        if rank == 0:
            nat.call("ttsk_comm_unique_id", uid)
        t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
        dist.broadcast(t, 0)
        uid = (ctypes.c_char * 128).from_buffer_copy(bytes(t.numpy().tobytes()))
        nat.call("ttsk_comm_init", uid, rank, world)

    B = max(1, int(args.batch))
    shape, cores, lcores, rcores = make_inputs(seed=3 + rank)
    _, _, lcores, rcores = (shape, cores) + tuple(make_inputs(seed=3)[2:])   # DRMs shared by all ranks
    all_cores = [cores] + [make_inputs(seed=3 + rank + 1000 * b)[1] for b in range(1, B)]   # B different TTs
    tts = [TensorTrain(c) for c in all_cores]
    tt = tts[0]
    left = TensorTrainDRM(L_RANK, shape, False, seed=1, cores=lcores)
    right = TensorTrainDRM(R_RANK, shape, True, seed=2, cores=rcores)
    plan = TTSketchPlan(tt.shape, tt.rank, left, right)
    inflight = max(1, min(int(args.inflight), nat.NUM_STREAMS // 2))
    from tt_sketch_amd.device import DevArray
    stride = plan.size + (plan.size & 1)            # even spacing keeps every sketch 16-byte aligned
    outs = [DevArray.empty((B * stride,)) for _ in range(inflight)]
    out = outs[0]
    sums = [DevArray.empty((plan.size,)) for _ in range(inflight)] if comm_on else None
    keep, flat = [], []
    for t in tts:
        p1, k1 = plan.core_pointers(t)
        keep.append(k1)
        flat += [p1[i] for i in range(plan.d)]
    ptrs = (ctypes.c_void_p * len(flat))(*flat)
    counter = [0]

    def run_on(slot):
        plan.run_batch(ptrs, B, outs[slot], stride, stream=2 * slot)   # stream pair (2 slot, 2 slot + 1)

    def step_eager():
        slot = counter[0] % inflight
        counter[0] += 1
        run_on(slot)
        if comm_on:
            # the B partial sketches of this rank are summed locally, then ONE all-reduce of one sketch
            # (32 MB) per step makes every rank hold the sketch of the world * B term sum.  All collectives go
            # together with the local sum to one dedicated stream in step order (no two collectives in flight on
            # different streams of one communicator); both overlap the products of step s + 1 on the other stream pair.
            cs = nat.NUM_STREAMS - 1
            nat.call("ttsk_stream_wait", cs, 2 * slot)       # the step's products are queued on stream 2 slot
            if plan.size % 2 == 0:
                nat.call("ttsk_sum_slices", ctypes.c_void_p(sums[slot].ptr), ctypes.c_void_p(outs[slot].ptr), B,
                         ctypes.c_size_t(stride), ctypes.c_size_t(plan.size), 0, cs)
            else:
                for b in range(B):
                    nat.call("ttsk_axpby", ctypes.c_void_p(sums[slot].ptr), ctypes.c_void_p(outs[slot].ptr + 8 * b * stride),
                             1.0, 1.0 if b else 0.0, ctypes.c_size_t(plan.size), cs)
            nat.call("ttsk_comm_allreduce_sum", ctypes.c_void_p(sums[slot].ptr), ctypes.c_size_t(plan.size), cs)
            # this slot's streams may touch outs[slot] / sums[slot] again only after its own sum + all-reduce; the
            # wait is queued now, so it names exactly this all-reduce and not the next step's
            nat.call("ttsk_stream_wait", 2 * slot, cs)

    for _ in range(inflight):
        step_eager()
    nat.call("ttsk_sync", -1)
    use_graph = bool(args.graph) and not comm_on
    graphs = []
    if use_graph:
        # one captured graph per in-flight slot: both chains' fork / join over the slot's stream pair
        for slot in range(inflight):
            g = ctypes.c_void_p()
            nat.call("ttsk_graph_begin", 2 * slot)
            run_on(slot)
            nat.call("ttsk_graph_end", 2 * slot, ctypes.byref(g))
            graphs.append(g)

    def step():
        if use_graph:
            slot = counter[0] % inflight
            counter[0] += 1
            nat.call("ttsk_graph_launch", graphs[slot], 2 * slot)
        else:
            step_eager()

    def barrier():
        nat.call("ttsk_sync", -1)
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    # latency of ONE sketch issued alone (batch 1, nothing else in flight), for reference
    single_ms, api_ms, cores_per_s_r53 = None, None, None
    if world == 1:
        one = (ctypes.c_void_p * plan.d)(*[ptrs[i] for i in range(plan.d)])
        for _ in range(3):
            plan.run(one, out)
        nat.call("ttsk_sync", -1)
        t1 = time.perf_counter()
        for _ in range(20):
            plan.run(one, out)
        nat.call("ttsk_sync", -1)
        single_ms = 1e3 * (time.perf_counter() - t1) / 20
        # T_total of SURVEY 8d = the reference's own timed region (scripts/experiment_base.py:102-113):
        # stream_sketch() through the Python API incl. DRM sampling, and with to_tt() on top
        import tt_sketch_amd as tsa
        api_ms = {}
        for name, fn in (("stream_sketch", lambda: tsa.stream_sketch(tt, left_rank=L_RANK, right_rank=R_RANK)),
                         ("stream_sketch_to_tt", lambda: tsa.stream_sketch(tt, left_rank=L_RANK, right_rank=R_RANK).to_tt())):
            best = float("inf")
            for it in range(8):
                nat.call("ttsk_sync", -1)
                t1 = time.perf_counter()
                fn()
                nat.call("ttsk_sync", -1)
                if it >= 2:
                    best = min(best, 1e3 * (time.perf_counter() - t1))
            api_ms[name] = best
        # the reference's other oversampling choice, right rank = left + 3 ("STTA+3", scripts/plot_timings.py:110-124;
        # SURVEY 8d asks for it next to r = 2 l): same batch, device-sampled DRM, 20 timed passes
        r53 = L_RANK + 3
        right53 = TensorTrainDRM(r53, shape, True, seed=2)
        plan53 = TTSketchPlan(tt.shape, tt.rank, left, right53)
        stride53 = plan53.size + (plan53.size & 1)
        out53 = DevArray.empty((B * stride53,))
        for _ in range(3):
            plan53.run_batch(ptrs, B, out53, stride53, stream=0)
        nat.call("ttsk_sync", -1)
        t1 = time.perf_counter()
        for _ in range(20):
            plan53.run_batch(ptrs, B, out53, stride53, stream=0)
        nat.call("ttsk_sync", -1)
        cores_per_s_r53 = D * B * 20 / (time.perf_counter() - t1)
        del out53, plan53, right53
        run_on(0)                      # restore the batched result checked below
        nat.call("ttsk_sync", -1)

    result = None
    if rank == 0:
        fl = algorithmic_flops(shape, (S_IN,) * (D - 1), (L_RANK,) * (D - 1), (R_RANK,) * (D - 1))
        # ---- roofline leg: per-launch device time of each GEMM class (hipEvents on the launch stream).
        # The pass runs the same sketches on ONE stream so that the bracketed times are not inflated
        # by the other chain's kernels sharing the CUs; they agree with rocprofv3's kernel durations.
        os.environ["TTSK_SINGLE_STREAM"] = "1"
        nat.call("ttsk_sync", -1)
        nat.call("ttsk_prof_enable", 1)
        reps = max(5, min(args.steps, 50))
        for _ in range(reps):
            run_on(0)
        nat.call("ttsk_sync", -1)
        classes = {}
        labels = {0: "right chain GEMM1  T = R^T X^T", 1: "right chain GEMM2  R' = sum T E (split-K slabs)",
                  2: "left chain GEMM1  T = L^T X", 3: "left chain GEMM2  L' = sum T D (split-K slabs)",
                  4: "Psi GEMM  Psi = T R", 5: "small products (Omega, first / last mode)"}
        for c, label in labels.items():
            n_l, ms, flops = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            nat.call("ttsk_prof_read", c, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(flops))
            kname = ctypes.create_string_buffer(96)
            nat.call("ttsk_prof_kernel_name", c, kname, 96)
            if n_l.value:
                classes[label] = dict(kernel=kname.value.decode() if c != 5 else "gemm_f64_kernel<...> (several)",
                                      launches_per_step=n_l.value / reps,
                                      avg_us=1e3 * ms.value / n_l.value,
                                      gflop_per_launch=flops.value / n_l.value * 1e-9,
                                      tflops=flops.value / (ms.value * 1e-3) * 1e-12 if ms.value else 0.0,
                                      share_ms=ms.value / reps)
        os.environ.pop("TTSK_SINGLE_STREAM", None)
        nat.call("ttsk_prof_enable", 0)
        # both roofs per class: algorithmic bytes of one launch (operands read once, result written once, the
        # shared DRM core once per launch) against HBM_TBS, flops against the matrix peak; the lower roof binds
        s_, l_, r_, n_ = S_IN, L_RANK, R_RANK, N_MODE
        mb = {labels[0]: B * (s_ * n_ * s_ + r_ * n_ * s_) + s_ * r_ * B,
              labels[1]: B * (r_ * n_ * s_ + s_ * r_) + r_ * n_ * r_,
              labels[2]: B * (s_ * n_ * s_ + l_ * n_ * s_) + s_ * l_ * B,
              labels[3]: B * (l_ * n_ * s_ + s_ * l_) + l_ * n_ * l_,
              labels[4]: B * (l_ * n_ * s_ + l_ * n_ * r_ + s_ * r_)}
        for label, doubles in mb.items():
            if label in classes:
                c = classes[label]
                c["algorithmic_mb_per_launch"] = doubles * 8e-6
                c["algorithmic_tb_per_s"] = doubles * 8 / (c["avg_us"] * 1e-6) * 1e-12
                hbm_roof_tf = c["gflop_per_launch"] * 1e9 / (doubles * 8) * HBM_TBS      # TF/s if bytes moved at HBM_TBS
                c["roof_tflops"] = min(PEAK_F64_MFMA_TF, hbm_roof_tf)
                c["bound"] = "hbm" if hbm_roof_tf < PEAK_F64_MFMA_TF else "mfma"
                c["frac_of_roof"] = c["tflops"] / c["roof_tflops"]
        dom = max(classes, key=lambda k: classes[k]["share_ms"])
        probe = ctypes.c_double()
        nat.call("ttsk_mfma_f64_peak_probe", ctypes.byref(probe))
        roofline = dict(bound="mfma", kernel=classes[dom]["kernel"], what=dom, achieved=classes[dom]["tflops"], peak=PEAK_F64_MFMA_TF,
                        unit="TFLOP/s", frac=classes[dom]["tflops"] / PEAK_F64_MFMA_TF,
                        traffic=load_traffic(classes[dom]["kernel"]),
                        avg_launch_us=classes[dom]["avg_us"], probed_mfma_f64_peak=probe.value,
                        classes=classes,
                        pipeline_tflops=fl["total"] * B * args.gpus * args.steps / elapsed * 1e-12)
        cpu = None
        parity = None
        if not args.no_cpu:
            cpu, ref = cpu_baseline(shape, cores, lcores, rcores)
            if world == 1:
                got = out.get()[:plan.size]
                want = np.concatenate([a.ravel() for a in ref[0] + ref[1]])
                parity = float(np.linalg.norm(got - want) / np.linalg.norm(want))
                if B > 1:   # the last tensor of the batch against the oracle as well
                    import oracle.ttsk_oracle as orc
                    ld, rd = orc.TTDrm(lcores, shape, False), orc.TTDrm(rcores, shape, True)
                    rP, rO = orc.general_sketch("tt", all_cores[-1], ld, rd, "streaming")
                    want = np.concatenate([a.ravel() for a in rP + rO])
                    got = out.get()[(B - 1) * stride:(B - 1) * stride + plan.size]
                    parity = max(parity, float(np.linalg.norm(got - want) / np.linalg.norm(want)))
        ms_step = 1e3 * elapsed / args.steps
        metric = "TT-cores sketched/sec (fp64) + achieved MFMA % for d=6 n=200 r=50 stream_sketch"
        try:
            with open(os.path.join(ROOT, "BASELINE.json")) as f:
                metric = json.load(f).get("metric", metric)
        except (OSError, ValueError):
            pass
        result = dict(metric=metric,
                      value=D * B * args.gpus * args.steps / elapsed, unit="TT-cores/s", n_gpus=args.gpus,
                      steps=args.steps, warmup=args.warmup, ms_per_step=ms_step, higher_is_better=True,
                      scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                      config=dict(workload="TensorTrain d=6 n=200 TT-rank 100, TensorTrainDRM left rank 50 / "
                                           "right rank 100, streaming sketch (both chains, Omega, Psi), "
                                           f"{B} TT(s) per GPU per step in one batched pass, {inflight} steps in flight" +
                                           ("; the rank's partial sketches are summed and ONE RCCL all-reduce of one sketch per step "
                                            "gives every rank the sketch of the whole sum" if world > 1 else ""),
                                  d=D, n=N_MODE, tt_rank=S_IN, left_rank=L_RANK, right_rank=R_RANK,
                                  algorithmic_gflop_per_sketch=fl["total"] * 1e-9, launch="hipGraph" if use_graph else "eager",
                                  tts_per_step=B, steps_in_flight=inflight, single_sketch_latency_ms=single_ms,
                                  t_total_ms_incl_drm_sampling=api_ms, tt_cores_per_s_right_rank_53=cores_per_s_r53,
                                  sketch_bytes=plan.size * 8),
                      roofline=roofline, cpu_baseline=cpu, parity_rel_err_vs_oracle=parity)
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
    if comm_on:
        nat.call("ttsk_comm_destroy")
    if dist is not None:
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
