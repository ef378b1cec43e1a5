"""Worker of tests/test_gpu_train.py::test_rccl_single_rank_engine_is_bit_identical (run as a child process so that the RCCL
communicator lives and dies with it): the data-parallel engine over a ONE-rank RCCL process group -- the real backend's
all-reduce calls, the two-piece overlapped exchange on the communication stream (as_artspeech_wait_head_grads) and the late
slice's exchange in the pipelined schedule -- must leave parameters, Adam moments and losses bit-identical to the engine
without a process group (a one-rank SUM is the identity)."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    from artspeech_amd.engine import TrainStep
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # "nccl" is RCCL on ROCm
    V, A, B, T = 45, 11, 32, 200
    lengths = torch.linspace(200, 60, 32).int()
    scale = 1.0 / (float(lengths.sum()) * A * 50)
    g = torch.Generator().manual_seed(3)
    batches = []
    for _ in range(4):
        x = torch.randint(1, V, (B, T), generator=g)
        tgt = torch.rand(B, T, A, 2, 50, generator=g)
        for b, l in enumerate(lengths):
            x[b, l:] = 0
            tgt[b, l:] = 0
        batches.append((x.to(dev), tgt.to(dev)))
    ld = lengths.to(dev)
    results = {}
    for pipeline in (False, True):
        for group in (None, dist.group.WORLD):
            torch.manual_seed(11)
            model = ArtSpeech(V, A).to(dev)
            step = TrainStep(model, B, T, lr=1e-3, weight_decay=1e-6, pipeline=pipeline, process_group=group)
            assert step.use_dist == (group is not None)
            assert step.ar_overlap == (group is not None), "the overlapped exchange is the RCCL path"
            losses = []
            for x, tgt in batches:
                step.step(x, ld, tgt, scale)
                losses.append(step.loss.clone())
            step.flush()
            torch.cuda.synchronize()
            results[(pipeline, group is not None)] = (model.flat.data.clone(), step.exp_avg.clone(), step.exp_avg_sq.clone(),
                                                      torch.stack(losses))
    ref = results[(False, False)]
    for key, got in results.items():
        for a, b, what in zip(ref, got, ("parameters", "exp_avg", "exp_avg_sq", "losses")):
            assert torch.equal(a, b), f"pipeline={key[0]} rccl={key[1]}: {what} differ, max |diff| {(a - b).abs().max().item():.3e}"
    assert torch.isfinite(ref[3]).all() and ref[3][-1] < ref[3][0]
    dist.destroy_process_group()
    print("rccl single rank ok")


if __name__ == "__main__":
    main()
