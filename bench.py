#!/usr/bin/env python3
"""bench.py -- ciphertexts/sec aggregated + PRE at N=2^16, L=12 (BASELINE.json metric).

One step = one pass of the hot path over one batch of synthetic client ciphertexts already resident in HBM:
    for each of C clients:  reencrypt(_accumulate)_batch: hybrid key-switch PRE into the common key
                            domain, folded coefficient-wise into the running aggregate            [SURVEY 8a a4+a5]
    (N>1 GPUs: RCCL reduce-scatter of the per-GPU partial sums as uint64 + reduce_mod)          [8e]
    rescale_mult_const(1/n_clients_total)  (EvalMult(ct, 1/n): rescale then integer constant)    [a6]
Unit of work = one client ciphertext PRE'd and folded into the aggregate (92.7 MB algorithmic).
Weak scaling: every GPU holds its own C clients x B ciphertexts; `value` = all ranks' units / max-over-ranks time.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 5 --warmup 2
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_MEASURED_GBS = 6290.0


def algorithmic_bytes_per_unit(N, L, K, beta, n_clients):
    """compulsory traffic per client ciphertext PRE'd and folded into the aggregate."""
    limb = 8 * N
    D = L + K
    pre = limb * (2 * L + 2 * beta * D + 2 * L)          # ct in + evk + ct out
    agg = limb * 2 * L * (1 + 1.0 / n_clients)            # read it once more, write 1/n of the sum
    resc = limb * (2 * L + 2 * (L - 1)) / n_clients       # rescale + const-mul of the sum, amortised
    return pre + agg + resc


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(args, log):
    """The CPU port (OpenMP over limbs like OpenFHE's WITH_OPENMP build) on a bounded sample of the same workload: `pre`
    ciphertexts PRE'd, summed, one rescale*const.  kind = "port".  ReEncrypt -- all of the work but one rescale per
    sample -- runs through oracle/cpu_fast.c: the restatement's algorithm carried out the way OpenFHE's native backend
    does (lazy Shoup butterflies, Barrett / Shoup constants, tables built once), word for word equal to the restatement
    (tests/test_oracle_algebra.py) and about twice as fast.  Timed twice: with ONE thread and with every core this
    process may use; `value` is the all-core number."""
    from oracle.oracle import FastCpuContext, set_threads
    t0 = time.time()
    o = FastCpuContext(args.log_n, args.depth, args.scaling_bits, 60, dnum=args.dnum)
    log(f"[cpu] oracle context built in {time.time() - t0:.1f}s")
    rng = np.random.default_rng(1)
    N, L, D = o.N, o.L, o.D

    def rnd(ids):
        out = np.empty((len(ids), N), dtype=np.uint64)
        for j, l in enumerate(ids):
            out[j] = rng.integers(0, int(o.moduli[l]), size=N, dtype=np.uint64)
        return out

    evk = rnd(list(range(D)) * (2 * o.beta)).reshape(o.beta, 2, D, N)
    pool = [rnd(list(range(L)) * 2).reshape(2, L, N) for _ in range(4)]

    def run(threads, n_pre):
        set_threads(threads)
        o.reencrypt(pool[0], evk)  # untimed warm-up (page faults, OpenMP team start)
        cts = [pool[i % 4] for i in range(n_pre)]
        f = o.const_factors(L - 1, 1, 1.0 / n_pre)
        t0 = time.time()
        acc = None
        for ct in cts:
            r = o.reencrypt(ct, evk)
            acc = r if acc is None else o.eval_add(acc, r)
        o.mult_factors(o.rescale(acc), f)  # the single rescale, amortised like the GPU step's (1 per n_clients units)
        return n_pre / (time.time() - t0), time.time() - t0

    # "all cores" = the cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box
    # shows every host core but grants one GPU's share) and by the 2L limb-level tasks the restatement has per call
    v1, dt1 = run(1, max(2, args.cpu_sample // 48))
    log(f"[cpu] 1 thread: {v1:.2f} ct/s ({dt1:.1f}s)")
    if "OMP_NUM_THREADS" in os.environ:
        n_all = int(os.environ["OMP_NUM_THREADS"])
    else:  # a box may show more cores than it grants: probe the candidates on a few ciphertexts, keep the fastest
        cands = sorted({min(usable_cores(), 2 * L), min(usable_cores(), 16), min(usable_cores(), 8)}, reverse=True)
        probe = {}
        for t in cands:
            probe[t] = run(t, 4)[0]
            log(f"[cpu] probe {t} threads: {probe[t]:.2f} ct/s")
            if probe[t] > 3.0 * v1:  # scales: no need to try fewer threads
                break
        n_all = max(probe, key=probe.get)
    va, dta = run(n_all, args.cpu_sample)
    log(f"[cpu] {n_all} threads: {va:.2f} ct/s ({dta:.1f}s)")
    return {"value": va, "unit": "ciphertexts/s", "cores": n_all, "kind": "port", "cpu_model": cpu_model(),
            "host_cores_visible": os.cpu_count(), "cores_usable": usable_cores(),
            "threads_1": {"value": v1, "cores": 1, "sample_ciphertexts": max(2, args.cpu_sample // 48), "seconds": dt1},
            "threads_all": {"value": va, "cores": n_all, "sample_ciphertexts": args.cpu_sample, "seconds": dta},
            "implementation": "oracle/cpu_fast.c (scalar C, 128-bit products, lazy Shoup butterflies; bit-equal to the "
                              "restatement oracle/mkckks_oracle.c, which is ~2x slower)",
            "sample": f"{args.cpu_sample} ciphertexts PRE'd + summed + 1 rescale*const at N=2^{args.log_n}, L={L}, "
                      f"dnum={args.dnum}; oracle/libcpufast.so (OpenMP over <= 2L limbs), all-core leg {dta:.1f}s, "
                      f"1-thread leg {dt1:.1f}s on {max(2, args.cpu_sample // 48)} ciphertexts"}


INT_BUTTERFLY_CYCLES = {False: 90.9, True: 77.6}  # Shoup | pseudo-Mersenne (profiles/r02_ubench_fpmod.txt)


def valu_ceiling(N, log_n, L, K, beta, alpha, n_clients, fp_limbs_q, int_pm=True):
    """Secondary ceiling: the path is 64-bit modular arithmetic on a chip without a 64-bit
    multiplier.  Work per unit x measured cycles per wave-operation (profiles/r02_ubench_intmul.txt, r02_ubench_fpmod.txt:
    fp64 butterfly 61, integer butterfly 90.9 (Shoup) or 77.6 (pseudo-Mersenne, the arithmetic mkckks_ctx_arith reports
    for the 60-bit limbs), v_mad_u64_u32 5.3, v_fma_f64 4.4 cycles) against 1024 SIMDs x 2.4 GHz."""
    D = L + K
    bf = N // 2 * log_n                                   # butterflies of one limb transform
    int_q = L - fp_limbs_q                                # integer-class Q limbs (60-bit q_0)
    # limb transforms of one PRE by arithmetic class: INTT c1, ModUp targets, INTT of the P limbs, ModDown targets
    t_int = int_q + (beta * K + (beta - 1) * int_q) + 2 * K + 2 * int_q
    t_fp = fp_limbs_q + (beta - 1) * fp_limbs_q + 2 * fp_limbs_q
    resc_int, resc_fp = (2 * int_q) / n_clients, (2 + 2 * (fp_limbs_q - 1)) / n_clients  # rescale, amortised
    conv_macs = (beta * alpha * (D - alpha) + 2 * K * L) * N   # ModUp + ModDown base-conversion MACs
    inner = 2 * beta * D * N                                # eval-key mul-adds
    cyc = ((t_int + resc_int) * bf * INT_BUTTERFLY_CYCLES[bool(int_pm)] + (t_fp + resc_fp) * bf * 61.0 + conv_macs * 4 * 5.3
           + inner * (fp_limbs_q / L * 6 * 4.4 + (1 - fp_limbs_q / L) * 30.0)) / 64.0
    return 1024 * 2.4e9 / cyc, {
        "limb_transforms_int": t_int, "limb_transforms_fp64": t_fp, "butterflies_per_limb": bf,
        "base_conv_macs": conv_macs, "inner_product_muladds": inner, "simd_cycles_per_unit": cyc,
        "int_butterfly": "pseudo-mersenne" if int_pm else "shoup", "int_butterfly_cycles": INT_BUTTERFLY_CYCLES[bool(int_pm)]}


def verify_exchange(torch, dist, ctx, pipe, agg, out, ct_in, evk, C, B, Bs, L, world, rank, inv_n, log):
    """Size-independent check of the N>1 exchange: the collective's unreduced integer sum + reduce_mod + rescale must
    equal eval_sum (modular) over the all-gathered per-rank aggregates + rescale, bit for bit."""
    torch.cuda.synchronize()
    last = (pipe["k"] - 1) & 1 if pipe is not None else 0
    got = (pipe["out"][last] if pipe is not None else out).clone()
    ctx.reencrypt_sum(ct_in, evk, agg, C, B, L)
    torch.cuda.synchronize()
    parts = torch.empty((world,) + tuple(agg.shape), dtype=agg.dtype, device=agg.device)
    if dist.get_backend() == "gloo":  # gloo moves GPU tensors only for broadcast / all_reduce
        parts.zero_()
        parts[rank].copy_(agg)
        dist.all_reduce(parts, op=dist.ReduceOp.SUM)
    else:
        dist.all_gather_into_tensor(parts, agg)
    total = torch.empty_like(agg)
    ctx.eval_sum(parts, total, world, B, L)
    want = torch.empty_like(got)
    ctx.rescale_mult_const(total[rank * Bs:(rank + 1) * Bs].contiguous(), want, Bs, L, inv_n)
    torch.cuda.synchronize()
    ok = torch.tensor([int(torch.equal(got, want))], device=agg.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok) != 1:
        raise SystemExit("[bench] --verify FAILED: exchange result differs from the modular sum")
    log("[bench] --verify ok: reduce-scatter + reduce_mod + rescale == modular sum of the per-rank aggregates")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--depth", type=int, default=10, help="mult depth; L = depth + 2 = 12")
    ap.add_argument("--scaling-bits", type=int, default=50)
    ap.add_argument("--dnum", type=int, default=3)
    ap.add_argument("--clients", type=int, default=8, help="clients per GPU")
    ap.add_argument("--cts", type=int, default=16, help="ciphertexts per client (multiple of --gpus)")
    ap.add_argument("--cpu-sample", type=int, default=1536,
                    help="ciphertexts in the CPU-baseline sample (about 17 s on 16 cores; the 1-thread leg takes 1/48 of it)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="repeat the timed block of --steps steps until this much of it has run; report the median block")
    ap.add_argument("--max-blocks", type=int, default=64)
    ap.add_argument("--verify", action="store_true",
                    help="N>1 only, after the timed region: check the integer reduce-scatter + reduce_mod + rescale "
                         "result against the modular sum of the gathered per-rank aggregates")
    ap.add_argument("--mode", choices=["sum", "accumulate"], default="sum",
                    help="sum: one reencrypt_sum_batch call over all clients (last ModDown pass + aggregation fused); "
                         "accumulate: per-client reencrypt_accumulate_batch calls spread over --streams")
    ap.add_argument("--shard", choices=["client", "ct"], default="client",
                    help="N>1: client = every GPU holds its own clients, one RCCL reduce-scatter of the partial sums "
                         "(SURVEY 8e.2); ct = every GPU holds ALL clients' re-encryption keys and a 1/N slice of the "
                         "ciphertext indices, no collective at all (SURVEY 8e.1)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the clients' PRE batches are spread over (each with its own context/workspace)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ppqsflhe_amd import Context
    from ppqsflhe_amd.sharding import reduce_partial_sums

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")

    def log(msg):
        if rank == 0:
            print(msg, file=sys.stderr, flush=True)

    # rehearsal knobs for a one-GPU box: every rank on device 0, collective carried by gloo
    backend = os.environ.get("MKCKKS_BENCH_BACKEND", "nccl")
    if os.environ.get("MKCKKS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
    if args.cts % world:
        raise SystemExit("--cts must be a multiple of --gpus")

    ctx = Context(args.log_n, args.depth, args.scaling_bits, 60, dnum=args.dnum, device=local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    # optional extra streams: memory-bound kernels (inner product, sums) of one client's batch overlap the
    # multiply-bound NTT kernels of another's; every stream has its own context (tables + workspace arena)
    side = []
    for _ in range(max(0, args.streams - 1) if args.mode == "accumulate" else 0):
        st = torch.cuda.Stream(device=dev)
        c2 = Context(args.log_n, args.depth, args.scaling_bits, 60, dnum=args.dnum, device=local_rank)
        c2.set_stream(st.cuda_stream)
        side.append((st, c2))
    N, L, K, D, beta = ctx.N, ctx.L, ctx.K, ctx.D, ctx.beta
    C, B = args.clients, args.cts
    by_ct = world > 1 and args.shard == "ct"
    if by_ct:  # this rank: every client of the job, B / world ciphertext indices -- the same C * B units per GPU
        if args.mode != "sum":
            raise SystemExit("--shard ct runs the reencrypt_sum path")
        C, B = args.clients * world, args.cts // world
    log(f"[bench] N=2^{args.log_n} L={L} K={K} dnum={args.dnum} beta={beta}; {C} clients x {B} ct per GPU, {world} GPU(s)"
        + (", sharded by ciphertext index (no collective)" if by_ct else ""))

    # synthetic inputs, generated in HBM: residues uniform in [0, q_i), seeded per rank
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)

    def uniform(shape_lead, ids):
        t = torch.empty(*shape_lead, len(ids), N, dtype=torch.int64, device=dev)
        for j, l in enumerate(ids):
            t[..., j, :] = torch.randint(0, int(ctx.moduli[l]), (*shape_lead, N), generator=gen, device=dev,
                                         dtype=torch.int64)
        return t

    ct_in = uniform((C, B), list(range(L)) * 2).view(C, B, 2, L, N)
    evk = uniform((C,), list(range(D)) * (2 * beta)).view(C, beta, 2, D, N)
    n_lanes = 1 + len(side)
    accs = torch.empty(n_lanes, B, 2, L, N, dtype=torch.int64, device=dev)  # one running aggregate per stream
    agg = torch.empty(B, 2, L, N, dtype=torch.int64, device=dev)
    exchange = world > 1 and not by_ct
    Bs = B // world if exchange else B
    shard = torch.empty(Bs, 2, L, N, dtype=torch.int64, device=dev) if exchange else agg
    out = torch.empty(Bs, 2, L - 1, N, dtype=torch.int64, device=dev)
    inv_n = 1.0 / (C * world) if exchange else 1.0 / C

    # the exchange itself: RCCL through the library's C-ABI (mkckks_reduce_scatter_sum_mod).  The communicator is
    # bootstrapped with 128 bytes from rank 0, carried by the torch.distributed group that also provides the barriers.
    # gloo rehearsals (every rank on one device: RCCL refuses duplicate GPUs) keep the torch.distributed collective.
    comms = {}

    def make_comm(cx):
        if not exchange or backend != "nccl" or os.environ.get("MKCKKS_BENCH_TORCH_COLLECTIVE") == "1":
            return None
        comm, err = None, ""
        try:
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(cx.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            comm = cx.comm_create(bytes(uid.cpu().numpy().tobytes()), world, rank)
        except Exception as e:  # keep the job alive on the torch.distributed collective
            err = str(e)
        # every rank must take the same route: one rank without a communicator sends all of them to torch.distributed
        ok = torch.tensor([1 if comm is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok) != 1:
            if comm is not None:
                cx.comm_destroy(comm)
            log(f"[bench] C-ABI communicator unavailable on some rank ({err or 'see the other ranks'}); "
                "using torch.distributed reduce_scatter")
            return None
        return comm

    def exchange_step(cx, partial, shard_out):
        comm = comms.get(id(cx))
        if comm is not None:
            cx.reduce_scatter_sum_mod(comm, partial, shard_out, Bs, L, world)
        else:
            reduce_partial_sums(partial, shard_out)
            cx.reduce_mod(shard_out, Bs, L, world)
    lanes = [(None, ctx)] + side

    # N>1: the exchange step of batch k (reduce-scatter over xGMI, reduce_mod, rescale of this rank's shard) runs on
    # its own stream with its own context while the main stream already key-switches batch k+1; partial sums,
    # shards and outputs are double-buffered, an event per buffer keeps batch k+2 off a buffer still in flight
    pipe = None
    if exchange and args.mode == "sum" and os.environ.get("MKCKKS_BENCH_SERIAL_EXCHANGE") != "1":
        comm = torch.cuda.Stream(device=dev)
        cx_tail = Context(args.log_n, args.depth, args.scaling_bits, 60, dnum=args.dnum, device=local_rank)
        cx_tail.set_stream(comm.cuda_stream)
        with torch.cuda.stream(comm):
            comms[id(cx_tail)] = make_comm(cx_tail)
        pipe = {"comm": comm, "ctx": cx_tail, "k": 0,
                "agg": [agg, torch.empty_like(agg)], "shard": [shard, torch.empty_like(shard)],
                "out": [out, torch.empty_like(out)], "free": [None, None]}

    def step_sum_pipelined():
        b = pipe["k"] & 1
        pipe["k"] += 1
        main = torch.cuda.current_stream()
        if pipe["free"][b] is not None:
            main.wait_event(pipe["free"][b])
        ctx.reencrypt_sum(ct_in, evk, pipe["agg"][b], C, B, L)
        ready = main.record_event()
        with torch.cuda.stream(pipe["comm"]):
            pipe["comm"].wait_event(ready)
            exchange_step(pipe["ctx"], pipe["agg"][b], pipe["shard"][b])
            pipe["ctx"].rescale_mult_const(pipe["shard"][b], pipe["out"][b], Bs, L, inv_n)
            pipe["free"][b] = pipe["comm"].record_event()

    def step_sum():
        if pipe is not None:
            return step_sum_pipelined()
        ctx.reencrypt_sum(ct_in, evk, agg, C, B, L)
        if exchange:
            exchange_step(ctx, agg, shard)
        ctx.rescale_mult_const(shard, out, Bs, L, inv_n)

    def step_accumulate():
        main = torch.cuda.current_stream()
        for st, _ in side:
            st.wait_stream(main)
        for c in range(C):
            lane = c % n_lanes
            cx = lanes[lane][1]
            if c < n_lanes:   # first client of the lane starts its aggregate
                cx.reencrypt(ct_in[c], evk[c], accs[lane], B, L)
            else:             # PRE folded straight into the running aggregate (no round trip through HBM)
                cx.reencrypt_accumulate(ct_in[c], evk[c], accs[lane], B, L)
        for st, _ in side:
            main.wait_stream(st)
        ctx.eval_sum(accs, agg, n_lanes, B, L)
        if exchange:
            # per-GPU partial sums are canonical (< 2^61): an integer sum over <= 8 ranks cannot wrap 2^64
            exchange_step(ctx, agg, shard)
        ctx.rescale_mult_const(shard, out, Bs, L, inv_n)

    step = step_sum if args.mode == "sum" else step_accumulate
    if exchange and pipe is None:
        comms[id(ctx)] = make_comm(ctx)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()

    def timed_block():
        """EXACTLY --steps steps between two fences (device synchronize + barrier on both sides); MAX over ranks of the
        wall time and of the HIP-event time on the stream the kernels run on."""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            step()
        ev1.record()
        fence()
        dt_b = time.perf_counter() - t0
        ms_b = ev0.elapsed_time(ev1)
        if world > 1:
            t = torch.tensor([dt_b, ms_b], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_b, ms_b = float(t[0]), float(t[1])
        return dt_b, ms_b

    # the timed block is repeated until about --min-seconds of it have run (a 20-step block is 0.1 s): the reported
    # figures are those of the MEDIAN block, with the fastest and the slowest beside them.  Every rank derives the same
    # repeat count from the first block's max-over-ranks time.
    blocks = [timed_block()]
    n_blocks = max(1, min(args.max_blocks, int(np.ceil(args.min_seconds / max(blocks[0][0], 1e-6)))))
    while len(blocks) < n_blocks:
        blocks.append(timed_block())
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
    dt, gpu_ms = blocks[order[(len(order) - 1) // 2]]  # lower median: an actual block, never an interpolation
    dt_min, dt_max = blocks[order[0]][0], blocks[order[-1]][0]

    if world > 1:
        gpu_ms = dt * 1e3  # the exchange runs on a second stream: the main-stream events miss it, wall time does not
        if args.verify and exchange:
            verify_exchange(torch, dist, ctx, pipe, agg, out, ct_in, evk, C, B, Bs, L, world, rank, inv_n, log)

    units_per_step = C * B * world
    value = units_per_step * args.steps / dt
    n_summed = C * world if exchange else C            # clients folded into one aggregate
    bytes_unit = algorithmic_bytes_per_unit(N, L, K, beta, n_summed)
    # one "launch" of the hot path = one step on one GPU (all of its kernels, serialised on one stream)
    step_s_gpu = gpu_ms / 1e3 / args.steps
    achieved = bytes_unit * C * B / step_s_gpu / 1e9
    traffic, traffic_source = None, None
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof) and world == 1:  # the PMC passes were taken on the single-GPU step
        try:
            rec = json.load(open(prof))
            key = f"logn{args.log_n}_L{L}_dnum{args.dnum}_C{C}_B{B}"
            if key in rec:
                traffic = rec[key].get("hbm_bytes_per_step")
                traffic_source = (f"profiles/hbm_traffic.json[{key}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a "
                                  f"builder run ({rec[key].get('profile', 'see profiles/')}), NOT measured in this run")
        except Exception:
            traffic = None
    fp_q = int(sum(1 for i in range(L) if int(ctx.arith[i]) == 1))
    int_pm = any(int(a) == 2 for a in ctx.arith)
    ceil_ct_s, ceil_terms = valu_ceiling(N, args.log_n, L, K, beta, ctx.alpha, n_summed, fp_q, int_pm)
    per_gpu = C * B / step_s_gpu

    shard_txt = ("" if world == 1 else
                 (f"sharded by ciphertext index x{world}, no collective -> " if by_ct else
                  "RCCL reduce_scatter(u64 sum)+reduce_mod via mkckks_reduce_scatter_sum_mod"
                  + (" (overlapped with the next batch's key switch)" if pipe else "") + " -> "))
    result = {
        "metric": f"ciphertexts/sec aggregated+PRE at N=2^{args.log_n}, L={L} RNS limbs",
        "value": value, "unit": "ciphertexts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "ms_per_step_min": dt_min * 1e3 / args.steps,
        "ms_per_step_max": dt_max * 1e3 / args.steps, "timed_blocks": len(blocks),
        "timing": f"median of {len(blocks)} timed blocks of {args.steps} steps each (fence on both sides of every block)",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"C3+C4: {C} clients x {B} ct per GPU, N=2^{args.log_n}, L={L}, K={K}, dnum={args.dnum}: "
                               + ("reencrypt_sum_batch (hybrid key-switch PRE of every client; row passes, inner product, "
                                  "ModDown tail and the sum over clients fused) -> " if args.mode == "sum" else
                                  "reencrypt_accumulate_batch (hybrid key-switch PRE folded into the aggregate) -> ")
                               + shard_txt + "rescale_mult_const(1/n)",
                   "ring_dim": N, "limbs": L, "special_limbs": K, "dnum": args.dnum, "clients_per_gpu": C,
                   "ct_per_client": B, "units_per_step": units_per_step,
                   "sharding": (f"ciphertext index x{world}" if by_ct else f"clients x{world}")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_unit": bytes_unit, "units_per_launch": C * B,
                     "launch": "one hot-path step on one GPU (all kernels of the path)",
                     "launch_ms": step_s_gpu * 1e3, "frac_of_measured_copy_peak": achieved / HBM_MEASURED_GBS,
                     "secondary": {"bound": "valu_int32_mul", "ceiling_ct_s": ceil_ct_s, "achieved_ct_s": per_gpu,
                                   "frac": per_gpu / ceil_ct_s, "terms": ceil_terms,
                                   "derivation": "gfx950 has no 64-bit multiplier: per unit, limb transforms x N/2 log2 N "
                                                 "butterflies x measured cycles per wave-operation (fp64-FMA butterfly 61, "
                                                 "integer butterfly on v_mad_u64_u32: 90.9 Shoup / 77.6 pseudo-Mersenne, "
                                                 "whichever mkckks_ctx_arith reports), base-conversion MACs x 4 "
                                                 "v_mad_u64_u32 x 5.3, eval-key mul-adds x (6 v_fma_f64 x 4.4 | 30), / 64 "
                                                 "lanes, against 1024 SIMDs x 2.4 GHz; cycle figures: "
                                                 "profiles/r02_ubench_intmul.txt, profiles/r02_ubench_fpmod.txt"}},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            result["cpu_baseline"] = cpu_baseline(args, log)
        except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU number
            result["cpu_baseline"] = {"value": None, "unit": "ciphertexts/s", "cores": 0, "kind": "port",
                                      "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    ctx.close()
    for _, c2 in side:
        c2.close()
    if pipe is not None:
        pipe["ctx"].close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
