"""Shared plumbing of the two reference-style entry points."""
import json
import os
import sys

# dmabuf IPC only on this host driver: the variable is read once, when the HIP runtime initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def init_distributed():
    """One process per GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the launcher (torchrun).  Single process otherwise
    -- like the reference's 'Not using distributed mode' (ALBEF_attack/utils.py:245-247)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("the attack path runs on MI355X GPUs only (no CPU fallback)")
    torch.cuda.set_device(local)
    # any torchrun launch initialises RCCL, also a 1-rank one: the success bits then go through the same device-tensor
    # all-gather as on 8 GPUs (run_sweep(force_collective=dist.is_initialized()))
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from vqattack_amd.whitebox import tuned_gemms
    tuned_gemms.enable()       # recorded library GEMM solutions for the white boxes (read-only; defaults when absent)
    return rank, world, torch.device("cuda", local)


def finish(rank, world, result, out_json=None):
    if rank == 0:
        print("acc_vqa", result["asr"], result["n_total"], flush=True)     # the reference's final print (vlmo_module.py:2122)
        print("sweep", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in result.items()
                                   if k in ("n_local", "seconds", "examples_per_sec_local", "gradient_steps", "n_batches",
                                            "mean_batch", "collectives")}), flush=True)
        if out_json:       # result["adv_text"] holds EVERY rank's samples (run_sweep gathers them): the complete output
            os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
            with open(out_json, "w") as f:
                json.dump(result["adv_text"], f)
    if dist.is_initialized():
        if rank == 0:
            print("dist_backend", dist.get_backend(), "world", dist.get_world_size(), "collectives",
                  result.get("collectives"), flush=True)
        dist.barrier()
        dist.destroy_process_group()
