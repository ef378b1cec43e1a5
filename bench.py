#!/usr/bin/env python
"""Benchmark of the phoneme_to_articulation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full training pass of the BiGRU encoder-decoder (BASELINE.json configs[1]) over one
synthetic batch: forward, masked Euclidean loss, backward to every parameter gradient, one flat RCCL
all-reduce of the gradients (N > 1) and the flat Adam update.  Inputs are resident in HBM before the
timed region.  Per rank: B=32 utterances x T=200 frames, A=11 articulators x 50 points, V=45, E=64,
H=128 (weak scaling: the global batch is 32*N, sharded by utterance).

Prints ONE JSON line on rank 0 (metric: articulator-frames/sec, fwd+bwd).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V, A, E, H, N = 45, 11, 64, 128, 50
B, T = 32, 200
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
# compulsory HBM bytes of one BiGRU fwd+bwd step at B=32, T=200 = 30.3 KB / frame
STEP_BYTES_PER_FRAME = 30.3e3


def gemm_flops(rows):
    """Algorithmic FLOPs of each named GEMM phase for `rows` = B*T frames (2 * M * N * K)."""
    D = 256
    f = {
        "head.gemm1": 2 * rows * A * D * H, "head.gemm2": 2 * rows * A * D * D, "head.gemm3": 2 * rows * A * 2 * N * D,
        "headb.dw3": 2 * rows * A * 2 * N * D, "headb.dx3": 2 * rows * A * 2 * N * D,
        "headb.dw2": 2 * rows * A * D * D, "headb.dx2": 2 * rows * A * D * D,
        "headb.dw1": 2 * rows * A * D * H, "headb.dx1": 2 * rows * A * D * H,
        "gru.xproj1": 2 * rows * 6 * H * 2 * H, "grub.dw_ih1": 2 * rows * 6 * H * 2 * H, "grub.dx1": 2 * rows * 6 * H * 2 * H,
        "grub.dw_hh": 2 * rows * 3 * H * H, "trunk.linear": 2 * rows * H * 2 * H, "trunkb.dw": 2 * rows * H * 2 * H,
        "trunkb.dx": 2 * rows * H * 2 * H,
    }
    return f


def phase_bytes(rows):
    """Algorithmic HBM bytes per launch of the non-GEMM phases (inputs read once + outputs written once)."""
    f4 = 4
    return {
        # recurrence, both directions: gi rows in, y + 4 gate planes out (forward); dy, y, gates in, dgi + dgh out
        "gru.fwd_l0": rows * 2 * (H + 4 * H) * f4, "gru.fwd_l1": rows * 2 * (3 * H + H + 4 * H) * f4,
        "gru.bwd_l1": rows * 2 * (H + H + 4 * H + 6 * H) * f4, "gru.bwd_l0": rows * 2 * (H + H + 4 * H + 6 * H) * f4,
        "loss": rows * A * 2 * N * 3 * f4,
    }


# phase name -> kernel name as rocprofv3 prints it (for the PMC traffic table committed under profiles/)
PMC_KERNEL = {"gru.bwd_l0": "gru_bwd_row_kernel<128>", "gru.bwd_l1": "gru_bwd_row_kernel<128>",
              "gru.fwd_l0": "gru_fwd_kernel<128, 4, true, true>", "gru.fwd_l1": "gru_fwd_kernel<128, 4, true, false>"}


def pmc_traffic(phase):
    """HBM bytes per launch of `phase`'s kernel from the committed rocprofv3 --pmc passes (separate
    FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied: tools/collect_profiles.py), or None."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if phase not in PMC_KERNEL or not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    for name, rec in table.items():
        if PMC_KERNEL[phase] in name:
            return rec["hbm_bytes_per_launch"]
    return None


def make_inputs(dev, seed):
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(1, V, (B, T), generator=g)
    targets = torch.rand(B, T, A, 2, N, generator=g)
    lengths = torch.full((B,), T, dtype=torch.int32)  # throughput runs: all lengths = T
    return tokens.to(dev), targets.to(dev), lengths


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the instrumented pass (roofline = null)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if os.environ.get("ARTSPEECH_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()  # rehearsal: ranks share the visible GPU(s)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm (one rank per GPU over xGMI); ARTSPEECH_DIST_BACKEND=gloo lets several ranks share ONE
        # GPU to rehearse the multi-rank code path on a single-GPU box (slow collectives, correctness only)
        backend = os.environ.get("ARTSPEECH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from artspeech_amd import _lib
    from artspeech_amd.engine import TrainStep
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech

    torch.manual_seed(0)  # same initial weights on every rank
    model = ArtSpeech(V, A, embed_dim=E, hidden_size=H, n_samples=N)
    state_dict = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    tokens, targets, lengths = make_inputs(dev, seed=1 + rank)  # each rank holds its own shard of utterances
    lengths_dev = lengths.to(dev)
    n_valid_global = int(lengths.sum()) * world
    scale = 1.0 / (n_valid_global * A * N)
    step = TrainStep(model, B, T, lr=1e-4, weight_decay=1e-6)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: warm-up {args.warmup} steps ...")
    for _ in range(args.warmup):
        step.step(tokens, lengths_dev, targets, scale)
    barrier()
    log("timed region ...")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.step(tokens, lengths_dev, targets, scale)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss_t = step.loss.detach().clone().reshape(1)
    if world > 1:
        dist.all_reduce(loss_t)  # shard losses are scaled by the global frame count: their SUM is the batch loss
    loss = float(loss_t.item())
    assert np.isfinite(loss), "loss is not finite"

    ms_per_step = 1e3 * elapsed / args.steps
    log(f"{ms_per_step:.3f} ms/step, loss {loss:.6f}")
    frames = B * T * world * args.steps
    value = frames / elapsed

    # ---- instrumented pass (rank 0): per-kernel-phase HIP events on the launch stream ----------------
    roofline, kernels = None, None
    if rank == 0 and not args.no_profile:
        L = _lib.lib()
        psteps = min(args.steps, 20)
        L.as_profile_reset()  # same configuration as the timed region (side-stream overlap on)
        L.as_profile_enable(1)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * psteps)]
        for i in range(psteps):
            step.forward_backward(tokens, lengths_dev, targets, scale)
        torch.cuda.synchronize()
        L.as_profile_enable(0)
        buf = C.create_string_buffer(1 << 16)
        L.as_profile_report(buf, len(buf))
        L.as_profile_reset()
        kernels = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            kernels[name] = {"launches_per_step": int(cnt) / psteps, "us_per_step": round(1e3 * float(ms) / psteps, 2)}
        rows = B * T
        flops, nbytes = gemm_flops(rows), phase_bytes(rows)
        # the two layers' backward recurrences are launches of ONE kernel: judge it as such (its rocprofv3 summary line under
        # profiles/ averages over both), then pick the kernel that costs most per step
        merged = dict(kernels)
        if "gru.bwd_l0" in merged and "gru.bwd_l1" in merged:
            a, b2 = merged.pop("gru.bwd_l0"), merged.pop("gru.bwd_l1")
            merged["gru.bwd_l1"] = {"launches_per_step": a["launches_per_step"] + b2["launches_per_step"],
                                    "us_per_step": a["us_per_step"] + b2["us_per_step"]}
        dom = max(merged, key=lambda k: merged[k]["us_per_step"])
        per_launch_us = merged[dom]["us_per_step"] / merged[dom]["launches_per_step"]
        if dom in flops:
            per_launch = flops[dom] / merged[dom]["launches_per_step"] if dom == "grub.dw_hh" else flops[dom]
            ach = per_launch / (per_launch_us * 1e-6) / 1e12
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                        "us_per_launch": round(per_launch_us, 2), "algorithmic_flops_per_launch": per_launch}
        else:
            per_launch = nbytes.get(dom, 0)
            ach = per_launch / (per_launch_us * 1e-6) / 1e9
            kname = "gru_bwd_row_kernel<128> (gru.bwd_l0 + gru.bwd_l1)" if dom == "gru.bwd_l1" else dom
            roofline = {"kernel": kname, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(dom), "us_per_launch": round(per_launch_us, 2),
                        "algorithmic_bytes_per_launch": per_launch,
                        "note": "dependent-step (latency) bound: 200 sequential recurrent steps per launch"}
        # whole-step view asked for by the north star: compulsory bytes of SURVEY 8(d) over the step time
        step_gbs = STEP_BYTES_PER_FRAME * B * T / (ms_per_step * 1e-3) / 1e9
        roofline["step_hbm"] = {"algorithmic_bytes_per_step": STEP_BYTES_PER_FRAME * B * T, "achieved_GBs": round(step_gbs, 1),
                                "frac_of_8TBs": round(step_gbs / HBM_PEAK_GBS, 5)}

    if rank == 0:
        result = {
            "metric": "articulator-frames/sec (fwd+bwd)", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "encoder_decoder BiGRU (ArtSpeech) V=45 E=64 H=128, 11 articulators x 50 pts, "
                                   "B=32 T=200 per GPU, all lengths 200; step = fwd + masked Euclidean loss + bwd + "
                                   "flat grad all-reduce + Adam",
                       "global_batch": B * world, "seq_len": T, "parallelism": f"dp{world}"},
            "loss": round(loss, 6),
            "roofline": roofline,
            "kernels_us_per_step": kernels,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(state_dict),  # rank 0 at N = 1 only
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
