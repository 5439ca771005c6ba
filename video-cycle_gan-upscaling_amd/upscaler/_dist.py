"""Data parallelism over frames: one process per GPU, torch.distributed (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference has no distributed code;
this is the north_star's DP extension: replicas of G and D, one flat-bucket all-reduce per model per
step (G <= 2.4 M, D <= 15.5 M fp32 elements), plus an all-reduce of the two D-output means so that
non-linear GAN losses see the global batch."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    return dist.group.WORLD


def allreduce_sum(flat, group):
    """in-place SUM over the ranks of one flat fp32 buffer (a model's whole gradient bucket, or the packed loss
    scalars).  The 1/ranks of the mean is folded into the consumer kernel (vcg_adam_keras_multi's grad_scale,
    vcg_gan_loss's mean_scale), so a data-parallel step adds one collective per bucket and no arithmetic pass."""
    if group is None or flat.numel() == 0:
        return flat
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # test rig only (several gloo ranks sharing one GPU, tests/test_dp_gpu.py): stage through the host
        host = flat.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class _Done:
    def wait(self):
        return None


def allreduce_sum_start(flat, group):
    """allreduce_sum without waiting: returns a handle whose wait() makes the current stream wait for the result.  With RCCL the
    collective runs on the communicator's own stream behind the work queued so far, so kernels launched between start and wait()
    overlap it -- they must not touch `flat`.  (gloo on device buffers, the test rig, completes at once through the host.)"""
    if group is None or flat.numel() == 0:
        return _Done()
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        allreduce_sum(flat, group)
        return _Done()
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)


def world_size(group):
    return 1 if group is None else dist.get_world_size(group)


def allreduce_mean(flat, group):
    """host-side helper (CPU tensors: tests, launch scripts): in-place mean over the ranks.  The device path of the
    trainer uses allreduce_sum and folds the division into its kernels."""
    if group is None or flat.numel() == 0:
        return flat
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # test rig only (several gloo ranks sharing one GPU, tests/test_dp_gpu.py): stage through the host
        host = flat.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host.div_(dist.get_world_size(group)))
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    return flat


def broadcast_(flat, group, src=0):
    if group is not None and flat.numel() > 0:        # (a model without BatchNormalization has an empty state buffer)
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            host = flat.detach().cpu()
            dist.broadcast(host, src=src, group=group)
            flat.copy_(host)
        else:
            dist.broadcast(flat, src=src, group=group)
    return flat


def shard_batch(global_batch, group):
    """per-rank slice [lo, hi) of a global batch of frames."""
    if group is None:
        return 0, global_batch
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    if global_batch % ws:
        raise ValueError("global batch %d not divisible by world size %d" % (global_batch, ws))
    per = global_batch // ws
    return rk * per, (rk + 1) * per
