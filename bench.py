#!/usr/bin/env python
"""Benchmark of the phoneme_to_articulation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--per-gpu-batch b]

With --gpus N > 1 and no torchrun environment, bench.py starts the N ranks itself as CHILD processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`)
before it touches the GPU, relays their output and exits with their code; under torchrun (RANK /
WORLD_SIZE set) it is one of the ranks.

One step = one full training pass of the BiGRU encoder-decoder (BASELINE.json configs[1]) over one
synthetic batch: forward, masked Euclidean loss, backward to every parameter gradient, one flat RCCL
all-reduce of the gradients (N > 1) and the flat Adam update.  Inputs are resident in HBM before the
timed region.  V=45, E=64, H=128, A=11 articulators x 50 points, T=200.
  weak scaling (default, the headline `value`): B=32 utterances per GPU, global batch 32*N;
  strong scaling (configs[2] "same as above, batch-sharded"): ONE global batch of 32 utterances dealt
  round-robin over the ranks (B=32/N per GPU), loss scaled by the global valid-frame count.
For N > 1 both are measured back to back and both are in the line (`scaling_runs`).

--per-gpu-batch b (N = 1 only): the same step on b utterances instead of 32 -- what ONE rank of an N = 32/b strong-scaling
run computes.  The default N = 1 run also measures b = 16, 8, 4 after the headline (`strong_scaling_bound`): the ceiling
T(32) / [T(32/N) + all-reduce] that configs[2] read as ONE global batch of 32 can reach, beside the weak reading.

Prints ONE JSON line on rank 0 (metric: articulator-frames/sec, fwd+bwd).  At N = 1 the line also
carries the driver-timed numbers of configs[3] (`transformer_c4`) and configs[4] on one GPU
(`pipeline_c5_1gpu`), the roofline of the dominant kernels and the CPU baseline.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

V, A, E, H, N = 45, 11, 64, 128, 50
B, T = 32, 200
D = 256                      # width of the ArticulatorPredictor hidden layers
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s specification (this box's own copy rate: roofline.peak_measured_copy)
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
# compulsory HBM bytes of one BiGRU fwd+bwd step at B=32, T=200 = 30.3 KB / frame
STEP_BYTES_PER_FRAME = 30.3e3
MAX_REHEARSAL_RANKS = 4      # gloo rehearsal (ranks share one card): the GPU box admits few processes per card


def gemm_flops(rows):
    """Algorithmic FLOPs per STEP of each named GEMM phase for `rows` = B*T frames (2 * M * N * K)."""
    return {
        "head.gemm1": 2 * rows * A * D * H, "head.gemm2": 2 * rows * A * D * D, "head.gemm3": 2 * rows * A * 2 * N * D,
        "headb.dw3": 2 * rows * A * 2 * N * D, "headb.dx3": 2 * rows * A * 2 * N * D,
        "headb.dw2": 2 * rows * A * D * D, "headb.dx2": 2 * rows * A * D * D,
        "headb.dw1": 2 * rows * A * D * H, "headb.dx1": 2 * rows * A * D * H,
        "gru.xproj1": 2 * rows * 6 * H * 2 * H, "grub.dw_ih1": 2 * rows * 6 * H * 2 * H, "grub.dx1": 2 * rows * 6 * H * 2 * H,
        "grub.dw_hh": 2 * 2 * rows * 2 * 3 * H * H,   # two layers x two directions
        "trunk.linear": 2 * rows * H * 2 * H, "trunkb.dw": 2 * rows * H * 2 * H, "trunkb.dx": 2 * rows * H * 2 * H,
        "gru.table0": 2 * V * 6 * H * E, "grub.dw_ih0": 2 * V * 6 * H * E, "grub.demb": 2 * V * 6 * H * E,
        # the head layers' and the trunk's weight gradients as ONE multi-problem launch (wgrad_f32.hip, as_wgrad_multi)
        "headb.dw_fused": 2 * rows * A * 2 * N * D + 2 * rows * A * D * D + 2 * rows * A * D * H + 2 * rows * H * 2 * H,
        "headb.dw31": 2 * rows * A * 2 * N * D + 2 * rows * A * D * H + 2 * rows * H * 2 * H,   # layers 3, 1 + trunk in one launch
    }


# GEMM phases grouped by the kernels that run them, named as rocprofv3 prints them (profiles/r02_kernel_stats.csv).
# head.gemm1/2 and headb.dx3/2 are the fused Linear + LayerNorm kernels: their time includes the LayerNorm epilogue.
GEMM_FAMILIES = {
    "forward (C = act(A.B^T + b)): lin_s6_kernel<1> | lin_f32_kernel<64, true, 1> [head Linear 1, 2 + ReLU + LayerNorm] + lin_out_s6_kernel | lin_out_kernel [output layer; its time includes the fused criterion epilogue] + gemm_f32_kernel<*, *, true, true, true>":
        ["gru.table0", "gru.xproj1", "trunk.linear", "head.gemm1", "head.gemm2", "head.gemm3"],
    "input gradients (C = A.B): lin_s6_kernel<2> | lin_f32_kernel<64, false, 2> [head dx 3, 2 + LayerNorm/ReLU backward] + lin_s6_plain_kernel<4> [heads dx1, split K] + gemm_f32_kernel<*, *, true, false, true>":
        ["headb.dx3", "headb.dx2", "headb.dx1", "trunkb.dx", "grub.dx1", "grub.demb"],
    "weight gradients (C = A^T.B): wgrad_f32_kernel<256, 32>(WgradMulti) [heads + trunk: two multi-problem launches, one of them a step late] + gemm_f32_kernel<64, 64, false, false, true> [GRU]":
        ["headb.dw_fused", "headb.dw31", "headb.dw3", "headb.dw2", "headb.dw1", "trunkb.dw", "grub.dw_ih1", "grub.dw_hh", "grub.dw_ih0"],
}


def phase_bytes(rows):
    """Algorithmic HBM bytes per launch of the non-GEMM phases (inputs read once + outputs written once)."""
    f4 = 4
    return {
        # recurrence, both directions: gi rows in, y + 4 gate planes out (forward); dy, y, gates in, dgi + dgh out
        "gru.fwd_l0": rows * 2 * (H + 4 * H) * f4, "gru.fwd_l1": rows * 2 * (3 * H + H + 4 * H) * f4,
        # layer 0 (token-table variant): no dgi; the per-utterance token sums [B][V][6H] leave the kernel instead
        "gru.bwd_l1": rows * 2 * (H + H + 4 * H + 6 * H) * f4, "gru.bwd_l0": rows * 2 * (H + H + 4 * H + 3 * H) * f4 + (rows // T) * V * 6 * H * f4,
        "loss": rows * A * 2 * N * 3 * f4,
    }


GRU_BWD_KERNEL = "gru_bwd_row_kernel<128, false, 2>"   # layer 1 (rocprofv3's name); layer 0 runs the token-sum variant
GRU_BWD_TOK_KERNEL = "gru_bwd_row_kernel<128, true, 2>"


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (separate FETCH_SIZE / WRITE_SIZE
    runs, gfx950 correction applied: tools/collect_profiles.py), newest round first, or None."""
    for tag in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            table = json.load(f)
        for name, rec in table.items():
            if kernel in name:
                return rec["hbm_bytes_per_launch"]
    return None


def pmc_step_traffic():
    """Whole-step HBM bytes (every kernel of this library, launches per step x PMC bytes per launch) from the newest committed
    rocprofv3 --pmc passes, or None: tools/collect_profiles.py writes `_step_traffic_bytes`."""
    for tag in ("r04",):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
        if os.path.exists(path):
            with open(path) as f:
                table = json.load(f)
            if "_step_traffic_bytes" in table:
                return int(table["_step_traffic_bytes"]), f"profiles/{tag}_pmc_traffic.json"
    return None, None


def measured_copy_gbs(dev, mib=1024, reps=64):
    """Streaming copy rate of THIS box (read + write bytes over time, a 1 GiB fp32 tensor copied `reps` times; the rate of
    the LAST quarter of the copies counts): the measured denominator SURVEY 8(d) asks for beside the 8 TB/s spec figure.

    A GPU that has been idle raises its clocks over its first ~20 ms of work (tools/step_warmup_curve.py: the step time of
    this benchmark falls smoothly from 1.23 to 1.10 ms over the first 20 steps of a process,
    profiles/r03_step_warmup_curve.txt), hence 64 copies with the last 16 counted.
    Returns (GB/s, milliseconds the probe kept the GPU busy)."""
    from artspeech_amd import _lib
    L = _lib.lib()
    src = torch.empty(mib << 18, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)

    def copy():   # the library's float4 grid-stride kernel (torch's copy_ reads 10-25 % lower on this box)
        _lib.check(L.as_copy_f32(_lib.ptr(src), _lib.ptr(dst), src.numel(), _lib.stream_ptr()), "as_copy_f32")
    copy()
    tail = max(1, reps // 4)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for i in range(reps):
        if i == reps - tail:
            e1.record()
        copy()
    e2.record()
    torch.cuda.synchronize()
    return round(2.0 * src.numel() * 4 * tail / (e1.elapsed_time(e2) * 1e-3) / 1e9, 1), round(e0.elapsed_time(e2), 2)


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(cores, 16)  # the GPU box's CPU share for one GPU


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(state_dict, seconds=12.0):
    """The CPU port (oracle/torch_port.py: the same graph on stock PyTorch CPU kernels, which is what the
    reference executes) timed on this box's host cores on the SAME workload shape (B=32, T=200)."""
    from oracle.torch_port import CpuPort
    cores = host_cores()
    torch.set_num_threads(cores)
    port = CpuPort(state_dict, A, H)
    g = torch.Generator().manual_seed(0)
    x = torch.randint(1, V, (B, T), generator=g)
    tgt = torch.rand(B, T, A, 2, N, generator=g)
    lengths = torch.full((B,), T, dtype=torch.int64)
    log(f"cpu baseline: {cores} threads, warm-up step ...")
    t_w = time.perf_counter()
    port.step(x, lengths, tgt)  # warm-up
    log(f"cpu baseline: warm-up step took {time.perf_counter() - t_w:.2f} s")
    n, t0 = 0, time.perf_counter()
    while True:
        port.step(x, lengths, tgt)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 50:
            break
    return {"value": round(n * B * T / el, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} fwd+bwd steps of the same B={B} T={T} A={A} batch on stock PyTorch CPU kernels "
                      f"(torch {torch.__version__}, {cores} threads), {el:.1f} s"}


def self_launch(args):
    """--gpus N > 1 outside torchrun: start the N ranks as child processes.  Nothing here initialises the GPU
    (device_count() only counts), and the parent never execs: it waits for the children and returns their code."""
    env = dict(os.environ)
    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        if ndev < 1:
            raise SystemExit("bench.py needs an MI355X")
        if args.gpus > MAX_REHEARSAL_RANKS:
            raise SystemExit(f"--gpus {args.gpus} but only {ndev} GPU(s) visible: a shared-card rehearsal is limited to "
                             f"{MAX_REHEARSAL_RANKS} ranks")
        # fewer cards than ranks: rehearse the multi-rank code path with the ranks sharing the visible card(s);
        # RCCL needs one device per rank, so the collectives go through gloo (correctness only, slow)
        env["ARTSPEECH_DIST_BACKEND"] = "gloo"
        log(f"{ndev} GPU(s) visible < --gpus {args.gpus}: REHEARSAL, ranks share the card, gloo collectives")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log("launching: " + " ".join(cmd))
    return subprocess.call(cmd, env=env)


def make_batch(n_utt, seed):
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(1, V, (n_utt, T), generator=g)
    targets = torch.rand(n_utt, T, A, 2, N, generator=g)
    lengths = torch.full((n_utt,), T, dtype=torch.int32)  # throughput runs: all lengths = T
    return tokens, targets, lengths


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="which run is the headline `value` (N > 1 measures both)")
    ap.add_argument("--per-gpu-batch", type=int, default=B,
                    help="N = 1 only: utterances in the batch (default 32); b < 32 = one rank's share of a strong-scaling run")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="every step self-contained (no weight-gradient work carried into the next step's forward)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the instrumented pass (roofline = null)")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[3] / configs[4] measurements (N = 1)")
    ap.add_argument("--no-exact", action="store_true", help="skip the second run on the exact fp32 matrix instruction (N = 1)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if B % world:
        raise SystemExit(f"--gpus {world} does not divide the global batch of {B} utterances")
    if args.per_gpu_batch != B and world > 1:
        raise SystemExit("--per-gpu-batch is a single-GPU measurement (no process group)")
    if args.per_gpu_batch < 1:
        raise SystemExit("--per-gpu-batch must be positive")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    backend = os.environ.get("ARTSPEECH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()  # rehearsal: ranks share the visible GPU(s)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm (one rank per GPU over xGMI); gloo = shared-card rehearsal (see self_launch)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from artspeech_amd import _lib
    from artspeech_amd.distributed import loss_scale, shard_batch
    from artspeech_amd.engine import TrainStep
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech

    torch.manual_seed(0)  # same initial weights on every rank
    model = ArtSpeech(V, A, embed_dim=E, hidden_size=H, n_samples=N)
    state_dict = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    flat0 = model.flat.data.clone()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(mode, b_one=None):
        """W warm-up + K timed steps of `mode`; returns (elapsed max over ranks, loss, TrainStep, inputs)."""
        model.flat.data.copy_(flat0)  # every run starts from the same parameters
        if mode == "weak":     # each rank holds its own batch of B utterances (b_one: --per-gpu-batch, N = 1)
            tokens, targets, lengths = make_batch(b_one or B, seed=1 + rank)
            n_valid_global = int(lengths.sum()) * world
        else:                  # one global batch of B utterances, dealt round-robin (artspeech_amd/distributed.py)
            tokens, targets, lengths = make_batch(B, seed=1)
            tokens, targets, lengths, n_valid_global = shard_batch(tokens, targets, lengths, rank, world)
        b_local = tokens.shape[0]
        tokens, targets = tokens.contiguous().to(dev), targets.contiguous().to(dev)
        lengths_dev = lengths.to(torch.int32).to(dev)
        scale = loss_scale(n_valid_global, A, N)
        # pipeline: the weight gradient / Adam update of the heads' second Linear of step i runs beside the forward
        # recurrences of step i + 1 (artspeech_amd/engine.py).  flush() applies what is pending: after it the parameters are
        # those of the unpipelined loop, bit for bit.  The timed region starts flushed and ends flushed: it holds the whole
        # work of exactly K steps.
        step = TrainStep(model, b_local, T, lr=1e-4, weight_decay=1e-6, pipeline=not args.no_pipeline)
        log(f"rank {rank}/{world} [{mode}]: B={b_local} per rank, warm-up {args.warmup} steps ...")
        for _ in range(args.warmup):
            step.step(tokens, lengths_dev, targets, scale)
        step.flush()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step.step(tokens, lengths_dev, targets, scale)
        step.flush()
        barrier()
        elapsed = time.perf_counter() - t0
        loss_t = step.loss.detach().clone().reshape(1)
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            dist.all_reduce(loss_t)  # shard losses are scaled by the global frame count: their SUM is the batch loss
        loss = float(loss_t.item())
        assert np.isfinite(loss), "loss is not finite"
        frames = b_local * T * world * args.steps
        rec = {"scaling": mode, "value": round(frames / elapsed, 1), "ms_per_step": round(1e3 * elapsed / args.steps, 4),
               "per_gpu_batch": b_local, "global_batch": b_local * world, "loss": round(loss, 6)}
        log(f"[{mode}] {rec['ms_per_step']:.3f} ms/step, {rec['value']:.0f} frames/s, loss {loss:.6f}")
        return rec, step, (tokens, lengths_dev, targets, scale)

    modes = [args.scaling] + ([m for m in ("weak", "strong") if m != args.scaling] if world > 1 else [])
    runs = {}
    head_step = head_inputs = None
    for m in modes:
        rec, st, inp = measure(m, args.per_gpu_batch if world == 1 else None)
        runs[m] = rec
        if m == args.scaling:
            head_step, head_inputs = st, inp
        else:
            del st, inp
    head = runs[args.scaling]
    ms_per_step = head["ms_per_step"]
    # N > 1: what the gradient exchange costs each rank, piece by piece (stream events, every rank runs the same steps)
    exchange = None
    if world > 1:
        st_, (tok_, len_, tgt_, sc_) = head_step, head_inputs
        st_.flush()
        st_.ar_timing = True
        for _ in range(min(args.steps, 10)):
            st_.step(tok_, len_, tgt_, sc_)
        st_.flush()
        st_.ar_timing = False
        mine = {"rank": rank, "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0, "ms": st_.ar_report()}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        exchange = {"note": "mean ms per step of each piece of the flat-gradient all-reduce on its own stream (tail: trunk + heads, beside "
                            "the GRU backward; head: GRU + embedding, after the backward; late: the deferred slice, in the next step's "
                            "forward) and `exposed`: the span of the compute stream between backward and Adam", "by_rank": gathered}
    # the same run on the exact fp32 matrix instruction everywhere (as_set_matrix_arith(0)): the headline's arithmetic is the
    # split one (fp32 operands as three bfloat16 planes, six plane products, fp32 accumulation) in the kernels that have it
    L_ = _lib.lib()
    arith_mode = int(L_.as_get_matrix_arith())
    exact_run = None
    if arith_mode != 0 and world == 1 and not args.no_exact:
        L_.as_set_matrix_arith(0)
        exact_run, st_, inp_ = measure(args.scaling, args.per_gpu_batch)
        del st_, inp_
        L_.as_set_matrix_arith(arith_mode)

    # ---- instrumented pass (rank 0): per-kernel-phase HIP events on each phase's own launch stream ----------------
    roofline, kernels = None, None
    if rank == 0 and not args.no_profile:
        step = head_step
        tokens, lengths_dev, targets, scale = head_inputs
        L = _lib.lib()
        psteps = min(args.steps, 20)
        L.as_profile_reset()  # same configuration as the timed region (side-stream overlap on)
        L.as_profile_enable(1)
        step.flush()
        use_dist, step.use_dist = step.use_dist, False   # rank 0 alone runs this pass: no collectives in it
        for i in range(psteps):   # whole steps: the pipelined schedule carries work across the step boundary
            step.step(tokens, lengths_dev, targets, scale)
        step.flush()
        step.use_dist = use_dist
        torch.cuda.synchronize()
        L.as_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        L.as_profile_report(buf, len(buf))
        L.as_profile_reset()
        kernels = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            kernels[name] = {"launches_per_step": int(cnt) / psteps, "us_per_step": round(1e3 * float(ms) / psteps, 2)}
        rows = head["per_gpu_batch"] * T
        flops, nbytes = gemm_flops(rows), phase_bytes(rows)
        # (1) the HBM-side entry: the backward recurrence, ONE kernel launched twice per step (layer 1, layer 0)
        gb = kernels.get("gru.bwd_l1")
        if gb:
            def entry(kernel, phase):
                us = kernels[phase]["us_per_step"] / kernels[phase]["launches_per_step"]
                ach = nbytes[phase] / (us * 1e-6) / 1e9
                return {"kernel": f"{kernel} ({phase})", "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(kernel),
                        "us_per_launch": round(us, 2), "algorithmic_bytes_per_launch": nbytes[phase]}
            # headline entry = the LONGER of the two backward-recurrence launches (layer 0 runs the token-sum variant)
            l1 = entry(GRU_BWD_KERNEL, "gru.bwd_l1")
            l0 = entry(GRU_BWD_TOK_KERNEL, "gru.bwd_l0") if ("gru.bwd_l0" in kernels and kernels["gru.bwd_l0"]["launches_per_step"] == 1) else None
            if l0 is not None and l0["us_per_launch"] > l1["us_per_launch"]:
                roofline, other, other_key = l0, l1, "layer1_variant"
            else:
                roofline, other, other_key = l1, l0, "layer0_variant"
            roofline["note"] = "dependent-step (latency) bound: 200 sequential recurrent steps per launch"
            roofline["traffic_source"] = "committed rocprofv3 --pmc passes of the same command (profiles/rNN_pmc_traffic.json), not this run"
            copy_gbs, _ = measured_copy_gbs(dev)
            roofline["peak_measured_copy"] = copy_gbs   # the box's own streaming-copy rate beside the 8 TB/s spec
            roofline["frac_of_measured_copy"] = round(roofline["achieved"] / copy_gbs, 5)
            if other is not None:
                roofline[other_key] = other
        else:
            roofline = {}
        # (2) the matrix side: every GEMM phase grouped by the kernel that runs it; FLOPs / time / fp32-MFMA peak.
        # The times are in-step (side-stream GEMMs share the chip with the recurrences), summed per family.
        fam, tot_f, tot_us = {}, 0.0, 0.0
        for name, phases in GEMM_FAMILIES.items():
            present = [p for p in phases if p in kernels]
            f = sum(flops[p] for p in present)
            us = sum(kernels[p]["us_per_step"] for p in present)
            if us <= 0:
                continue
            tf = f / (us * 1e-6) / 1e12
            fam[name] = {"flops_per_step": f, "us_per_step": round(us, 1), "achieved": round(tf, 1), "unit": "TFLOP/s",
                         "peak": F32_MFMA_PEAK_TFLOPS, "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 3),
                         "phases": {p: round(flops[p] / (kernels[p]["us_per_step"] * 1e-6) / 1e12, 1) for p in present}}
            tot_f += f
            tot_us += us
        top = max(fam, key=lambda k: fam[k]["us_per_step"]) if fam else None
        roofline["gemm"] = {"bound": "mfma", "top_family": top, "families": fam,
                            "all": {"flops_per_step": tot_f, "us_per_step": round(tot_us, 1),
                                    "achieved": round(tot_f / (tot_us * 1e-6) / 1e12, 1) if tot_us else None,
                                    "unit": "TFLOP/s", "peak": F32_MFMA_PEAK_TFLOPS,
                                    "frac": round(tot_f / (tot_us * 1e-6) / 1e12 / F32_MFMA_PEAK_TFLOPS, 3) if tot_us else None}}
        # (3) whole-step view asked for by the north star: compulsory bytes of SURVEY 8(d) over the step time
        step_gbs = STEP_BYTES_PER_FRAME * rows / (ms_per_step * 1e-3) / 1e9
        roofline["step_hbm"] = {"algorithmic_bytes_per_step": STEP_BYTES_PER_FRAME * rows, "achieved_GBs": round(step_gbs, 1),
                                "frac_of_8TBs": round(step_gbs / HBM_PEAK_GBS, 5)}
        # what the step REALLY moves through HBM (PMC counters of every kernel, committed profile) against the compulsory bytes
        st_bytes, st_src = pmc_step_traffic()
        roofline["step_traffic_bytes"] = st_bytes
        roofline["step_traffic_over_compulsory"] = round(st_bytes / (STEP_BYTES_PER_FRAME * B * T), 2) if st_bytes else None
        roofline["step_traffic_source"] = st_src
    del head_step, head_inputs

    if rank == 0:
        extras = {}
        if world == 1 and not args.no_extras and args.per_gpu_batch == B:
            # what one rank of a strong-scaling run of configs[2] computes: the same step on 32/N utterances
            t32 = ms_per_step
            sweep = {}
            for n_ranks in (2, 4, 8):
                rec, st, inp = measure("weak", B // n_ranks)
                del st, inp
                tb = rec["ms_per_step"]
                # gradient all-reduce of 7.46 MB, NOT measured here (one GPU): a ring over xGMI moves 2(N-1)/N of the
                # buffer per rank; assumed 100 GB/s effective + 30 us latency; the two-piece schedule hides 74 % of the
                # buffer beside the GRU backward, so the exposed time lies between the two figures
                ar_ms = (2.0 * (n_ranks - 1) / n_ranks * 7.46e6 / 100e9 + 30e-6) * 1e3
                sweep[f"N={n_ranks}"] = {"per_gpu_batch": B // n_ranks, "ms_per_step_one_rank": tb,
                                         "speedup_ceiling_allreduce_hidden": round(t32 / tb, 2),
                                         "speedup_allreduce_exposed": round(t32 / (tb + ar_ms), 2),
                                         "assumed_allreduce_ms": round(ar_ms, 3)}
            extras["strong_scaling_bound"] = {
                "note": "configs[2] read as ONE global batch of 32 dealt over N ranks: T(32) / [T(32/N) + all-reduce], T measured "
                        "on this GPU, all-reduce ASSUMED (100 GB/s effective ring + 30 us); the recurrences' 800 dependent "
                        "steps do not shrink with the batch, so >= 6x at N = 8 is only reachable under weak scaling",
                "ms_per_step_B32": t32, "by_ranks": sweep}
        if world == 1 and not args.no_extras:
            # configs[3] / configs[4] on this GPU, timed by the same command (their own workloads, not `value`)
            torch.cuda.empty_cache()
            try:
                import bench_pipeline
                import bench_transformer
                t_ = bench_transformer.run(32, 200, 3, dev, log)
                c4 = dict(t_["fwd_bwd"])
                from artspeech_amd.phoneme_to_articulation.transformer import ops as t_ops
                names = {v: k for k, v in t_ops.PRECISIONS.items()}
                c4.update({"workload": "ArtSpeechTransformer d=256 L=6 heads=4 A=11, B=32 T=200, fwd + masked loss + bwd, fp32",
                           "arith": (f"forward GEMMs: {names[t_ops.GEMM_PRECISION]}; backward GEMMs (input and weight gradients): "
                                     f"{names[t_ops.GRAD_PRECISION]} (lib = the library's matrix arithmetic, see `arith`); attention, "
                                     "LayerNorm, softmax: fp32"),
                           "params": t_["params"], "fwd_only": t_["fwd"]})
                extras["transformer_c4"] = c4
                p_ = bench_pipeline.run(32, 200, 3, dev, log)
                p_["workload"] = ("transformer forward -> tract variables -> area function + resampling -> DeepSpeech2 scorer "
                                  "top-1, B=32 T=200, one GPU (configs[4] shards utterances over 8)")
                extras["pipeline_c5_1gpu"] = p_
                # the three small kernels north_star names besides the recurrence, alone, HIP-event timed: GB/s vs 8 TB/s
                import bench_metrics_kernels
                extras["metrics_kernels"] = bench_metrics_kernels.main(iters=30, log=log)
            except Exception as exc:  # the headline must survive a failure of the extras
                extras["extras_error"] = f"{type(exc).__name__}: {exc}"
                log(f"extras failed: {extras['extras_error']}")
        result = {
            "metric": "articulator-frames/sec (fwd+bwd)", "value": head["value"], "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"encoder_decoder BiGRU (ArtSpeech) V=45 E=64 H=128, 11 articulators x 50 pts, T=200, "
                                   f"B={head['per_gpu_batch']} per GPU ({args.scaling} scaling), all lengths 200; step = fwd + "
                                   "masked Euclidean loss + bwd + flat grad all-reduce + Adam" + ("" if args.no_pipeline else
                                   "; software-pipelined across the step boundary (one weight gradient + its Adam slice run beside the "
                                   "next step's forward recurrences; flushed inside the timed region)"),
                       "global_batch": head["global_batch"], "seq_len": T, "parallelism": f"dp{world}"},
            "loss": head["loss"],
            # fp32 in, fp32 out, fp32 accumulation everywhere; how the PRODUCTS of the matrix kernels are formed:
            "arith": ("bf16x6: every fp32 operand split exactly into three bfloat16 numbers, six of the nine plane products on "
                      "v_mfma_f32_32x32x16_bf16, fp32 accumulate -- in the fused head layers (Linear + LayerNorm forward and backward), "
                      "the output layer and the heads' input gradient; v_mfma_f32_32x32x2_f32 (exact fp32) in the GRU-side GEMMs and in "
                      "the weight gradients that run beside the recurrences; error against fp64 <= the fp32 instruction's "
                      "(tests/test_gpu_parity.py::test_split_matrix_arithmetic_*)") if arith_mode else "fp32: v_mfma_f32_32x32x2_f32 everywhere",
            "exact_fp32": ({"ms_per_step": exact_run["ms_per_step"], "value": exact_run["value"], "loss": exact_run["loss"],
                            "note": "same command, as_set_matrix_arith(0): every matrix kernel on v_mfma_f32_32x32x2_f32"}
                           if exact_run else None),
            # ranks that exchanged gradients over RCCL (0 in a gloo rehearsal, where no RCCL communicator exists)
            "rccl_ranks": (dist.get_world_size() if (world > 1 and dist.get_backend() == "nccl") else (1 if world == 1 else 0)),
            "dist_backend": (dist.get_backend() if world > 1 else None),
            "rehearsal_shared_gpu": bool(world > 1 and backend != "nccl"),
            "scaling_runs": runs,
            "gradient_exchange": exchange,
            "roofline": roofline,
            "kernels_us_per_step": kernels,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(state_dict),  # rank 0 at N = 1 only
        }
        result.update(extras)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
