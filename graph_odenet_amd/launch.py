"""Start-up of the multi-GPU benchmarks (bench.py, tools/qc_bench.py, tools/partition_bench.py): one process per GPU.

Two ways in, one state out:
  * launched by `python -m torch.distributed.run --nproc-per-node N ... script --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE
    in the environment): `init_ranks` joins the process group;
  * launched bare as `python script --gpus N`: `self_launch` starts the N ranks itself as FRESH child processes through
    torch.distributed.run, BEFORE this process has made any GPU call (a process that has initialised the GPU must neither
    exec nor fork GPU work), relays their output and returns the children's exit code.  Importing torch and
    `torch.cuda.device_count()` do not initialise the GPU on this image.

Backend: "nccl" (= RCCL over xGMI) when the node has a GPU per rank.  With fewer GPUs than ranks - the rehearsal on a
one-GPU box - the ranks share the GPUs (device = local_rank mod device_count) and exchange through gloo staged through
the host (parallel.py / partition.py handle that case); RCCL cannot place two ranks on one device.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def needs_self_launch(n_gpus):
    return n_gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ


def self_launch(script, argv, n_gpus):
    """Run `script argv` as n_gpus ranks of one node; returns the exit code of torch.distributed.run.  Call this before
    anything touches the GPU; never replaces the running process."""
    import torch
    if torch.cuda.is_initialized():
        raise RuntimeError("self_launch: the GPU is already initialised in this process; start the ranks first")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n_gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(script)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def init_ranks(backend="auto", timeout_s=600):
    """Returns (rank, local_rank, world, device, backend).  Initialises the process group when WORLD_SIZE > 1 and runs
    one checked all-reduce of 1.0 (the sum must equal the number of ranks)."""
    import datetime
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise RuntimeError("no GPU visible")
    if backend == "auto":
        backend = "nccl" if n_dev >= world else "gloo"
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        to = datetime.timedelta(seconds=timeout_s)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=to)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=to)
        one = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        if int(one.item()) != world:
            raise RuntimeError("all-reduce of 1.0 over %d ranks gave %r" % (world, one.item()))
    return rank, local_rank, world, dev, backend
