"""One process per GPU (train.py:58-64, frozen_ppo.py:116-126): rank / world from the launcher's environment,
device = the local rank's GPU, process group over RCCL ("nccl" is RCCL on ROCm).

``IGI_DIST_BACKEND=gloo`` keeps the whole multi-rank control flow testable on a single-GPU box: ranks share the
devices round-robin and the collectives run through gloo (which carries device tensors)."""
import os

# dmabuf IPC is the only mode the host driver supports; the ROCm runtime reads this when HSA initialises, i.e. at the
# first GPU call of the process -- so it is set at import, before anything here (or after it) touches a device.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def init_rank_device():
    """-> (local_rank, world_size, device string); initialises the default process group once."""
    rank = int(os.getenv("LOCAL_RANK", "0"))
    world = int(os.getenv("WORLD_SIZE", "1"))
    backend = os.environ.get("IGI_DIST_BACKEND", "nccl")
    index = rank if backend == "nccl" else rank % max(torch.cuda.device_count(), 1)
    device = "cuda:" + str(index)
    torch.cuda.set_device(index)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", rank=int(os.getenv("RANK", rank)), world_size=world,
                                    device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=int(os.getenv("RANK", rank)), world_size=world)
    return rank, world, device
