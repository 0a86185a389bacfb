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
        import datetime
        limit = datetime.timedelta(seconds=float(os.environ.get("VQA_DIST_TIMEOUT", "600")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=limit)
    from vqattack_amd.whitebox import tuned_gemms
    tuned_gemms.enable()       # recorded library GEMM solutions for the white boxes (read-only; defaults when absent)
    return rank, world, torch.device("cuda", local)


def seed_everything(seed, rank):
    """``seed = args.seed + rank`` into torch / numpy / random (ALBEF_attack/VQA.py:74-77): the random start of a sample's
    first PGD block (``time == 0``) and ALBEF's per-forward token masking draw from torch's generators."""
    import random
    import numpy as np
    torch.manual_seed(seed + rank)
    np.random.seed(seed + rank)
    random.seed(seed + rank)


def file_source(flavor, questions, image_root, text_len, image_size, vocab_file="", tables_dir="", joint=True):
    """``VqaFilePairs`` over the reference's input files: VQA annotation json(s) (``test_file``), the image directory
    (``vqa_root``), a BERT ``vocab.txt`` for textual questions, and the directory holding the in-tree tables
    (``right_part*.txt``, ``*_ans_table*.txt``, ``chatgpt_all_5k*.txt``, ``all_correct_ans*.txt``)."""
    from vqattack_amd.attack.dataset import VqaFilePairs, load_tables
    from vqattack_amd.attack.wordpiece import WordPiece
    tok = WordPiece(vocab_file) if vocab_file else None
    tables = load_tables(tables_dir, flavor) if tables_dir else None
    return VqaFilePairs(questions, image_root, flavor, text_len, image_size, tokenizer=tok, tables=tables, joint=joint)


def mlm_proposer(checkpoint, vocab_file, device):
    """The reference's candidate proposer, ``BertForMaskedLM.from_pretrained('bert-base-uncased')`` (adv_attack.py:110),
    from a local HF state dict, + the ids it may never propose (``##`` pieces and stop words, adv_attack.py:253-258).
    Returns ``(mlm_logits_fn, banned_ids)``; ``(None, None)`` without a checkpoint (the white box's own MLM head then
    proposes)."""
    if not checkpoint:
        return None, None
    from vqattack_amd.attack.dataset import DEFAULT_STOP_WORDS
    from vqattack_amd.attack.proposer import BertMlmProposer, banned_ids
    from vqattack_amd.attack.wordpiece import WordPiece
    model = BertMlmProposer.from_hf_state_dict(load_checkpoint(checkpoint)).to(device)
    banned = banned_ids(WordPiece(vocab_file).tokens, DEFAULT_STOP_WORDS).to(device) if vocab_file else None
    return model, banned


def load_checkpoint(path):
    """``torch.load(path, map_location='cpu')`` of a reference checkpoint (adv_attack.py:83,96; vlmo_module.py:690) --
    tensors only (``weights_only``): a checkpoint is data, never code."""
    return torch.load(path, map_location="cpu", weights_only=True)


def finish(rank, world, result, out_json=None):
    if rank == 0:
        print("acc_vqa", result["asr"], result["n_total"], flush=True)     # the reference's final print (vlmo_module.py:2122)
        print("sweep", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in result.items()
                                   if k in ("n_local", "seconds", "examples_per_sec_local", "gradient_steps", "n_batches",
                                            "mean_batch", "collectives", "input_seconds", "input_blocked_seconds",
                                            "writer_seconds", "skipped")}), flush=True)
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
