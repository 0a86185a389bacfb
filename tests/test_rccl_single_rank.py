"""The RCCL branch of the two launch paths, on the one GPU of the test box.

The driver's multi-GPU runs start ``bench.py`` / ``entry/run.py`` under ``torch.distributed.run`` with one rank per GPU and
``backend="nccl"`` (= RCCL on ROCm); the 2-rank rehearsal (``test_bench_multirank.py``) has to use gloo because two ranks
share one card.  Here ONE rank is launched exactly like a torchrun rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment, no ``VQA_DIST_BACKEND`` override): ``init_process_group("nccl", device_id=...)`` runs, the success ledger's
all-gathers run on DEVICE tensors through RCCL (``force_collective``), the barrier and ``destroy_process_group`` run.
The rank is forked from the pre-GPU fork server of ``tests/conftest.py`` (no exec from a GPU-initialised process); a
failing rank exits non-zero.
"""
import json
import multiprocessing
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _as_torchrun_rank(port, out_path):
    os.environ.pop("VQA_DIST_BACKEND", None)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    out = open(out_path, "w")
    os.dup2(out.fileno(), 1)
    os.dup2(out.fileno(), 2)


def _bench_rank(port, out_path, argv):
    _as_torchrun_rank(port, out_path)
    sys.argv = ["bench.py"] + list(argv)
    import bench
    bench.main()
    sys.stdout.flush()


def _entry_rank(port, out_path, argv):
    _as_torchrun_rank(port, out_path)
    sys.path.insert(0, os.path.join(ROOT, "entry"))
    sys.argv = ["run.py"] + list(argv)
    import run
    run.main()
    sys.stdout.flush()


def _entry_vqa_rank(port, out_path, argv):
    _as_torchrun_rank(port, out_path)
    sys.path.insert(0, os.path.join(ROOT, "entry"))
    sys.argv = ["VQA.py"] + list(argv)
    import VQA
    VQA.main()
    sys.stdout.flush()


def _run(target, args):
    ctx = multiprocessing.get_context("forkserver")
    p = ctx.Process(target=target, args=args)
    p.start()
    p.join(timeout=420)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("the rank did not finish within 420 s")
    return p.exitcode


def test_bench_single_rank_over_rccl(tmp_path):
    out = str(tmp_path / "bench.out")
    argv = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--model", "vlmo_tiny", "--batch", "4", "--pgd-steps", "6",
            "--no-cpu-baseline", "--no-b256"]
    code = _run(_bench_rank, (_free_port(), out, argv))
    text = open(out).read()
    assert code == 0, text[-2000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 2
    coll = rec["collective"]
    assert coll is not None and coll["backend"] == "nccl" and coll["world"] == 1
    assert coll["tensor_device"].startswith("cuda") and coll["calls"] == 1      # ONE all-gather of the bits, on a device tensor
    assert rec["attack_success_rate"] is not None
    assert rec["distinct_devices"] == 1 and rec["device_rank0"]         # gathered through RCCL as two int64 words


def test_bench_sweep_single_rank_over_rccl(tmp_path):
    """``bench.py --sweep`` (BASELINE configs[3]'s form) as a 1-rank torchrun job: the sweep's two all-gathers (success bits,
    adversarial text) and the per-rank record / device-identity gathers run on DEVICE tensors through RCCL; under RCCL the
    number of distinct devices must equal the number of ranks (here 1 == 1), and the line names rank 0's GPU."""
    out = str(tmp_path / "bench_sweep.out")
    argv = ["--gpus", "1", "--sweep", "40", "--warmup", "1", "--model", "vlmo_tiny", "--batch", "16", "--pgd-steps", "8"]
    code = _run(_bench_rank, (_free_port(), out, argv))
    text = open(out).read()
    assert code == 0, text[-2000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    rec = json.loads(lines[0])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 1 and rec["config"]["n_samples"] == 40
    assert rec["collective"] == {"backend": "nccl", "world": 1, "calls": 2}
    assert rec["distinct_devices"] == 1 and rec["device_rank0"] and ("uuid:" in rec["device_rank0"] or "pci:" in rec["device_rank0"])
    assert rec["per_rank"]["samples"] == [40] and rec["per_rank"]["n_batches"] == [3] and rec["value"] > 0


def test_entry_run_single_rank_over_rccl(tmp_path):
    out = str(tmp_path / "run.out")
    code = _run(_entry_rank, (_free_port(), out, ["with", "tiny", "n_samples=6", "per_gpu_batchsize=3", "dual_every=3",
                                                   "mixed=True", "attack_dir=" + str(tmp_path / "attack_dir_VLMO")]))
    text = open(out).read()
    assert code == 0, text[-2000:]
    lines = text.splitlines()
    acc = [ln for ln in lines if ln.startswith("acc_vqa")]
    assert len(acc) == 1 and acc[0].split()[2] == "6"
    info = [ln for ln in lines if ln.startswith("dist_backend")]
    assert len(info) == 1, text[-2000:]
    adv = json.load(open(str(tmp_path / "attack_dir_VLMO" / "adv_txt.json")))      # vlmo_module.py:2095-2097
    assert sorted(map(int, adv)) == list(range(6)) and all(len(v) == 8 for v in adv.values())
    parts = info[0].split()
    assert parts[1] == "nccl" and parts[3] == "1" and int(parts[5]) == 2       # success bits + adversarial text


def test_entry_vqa_single_rank_over_rccl(tmp_path):
    """The ALBEF-flavor entry point (argparse + yaml like ALBEF_attack/VQA.py:119-134) through the same launch path, with
    a mixed sweep (feature and dual-loss samples of different schedules in one batch) and the .pt / json outputs."""
    out = str(tmp_path / "vqa.out")
    code = _run(_entry_vqa_rank, (_free_port(), out, ["--tiny", "--n_samples", "6", "--dual_every", "3", "--mixed",
                                                       "--output_dir", str(tmp_path)]))
    text = open(out).read()
    assert code == 0, text[-2000:]
    lines = text.splitlines()
    acc = [ln for ln in lines if ln.startswith("acc_vqa")]
    assert len(acc) == 1 and acc[0].split()[2] == "6"
    info = [ln for ln in lines if ln.startswith("dist_backend")]
    assert len(info) == 1 and info[0].split()[1] == "nccl", text[-2000:]
    sweep = json.loads([ln for ln in lines if ln.startswith("sweep ")][0][len("sweep "):])
    assert sweep["n_local"] == 6 and sweep["n_batches"] >= 1 and sweep["collectives"] >= 2
    adv = json.load(open(os.path.join(str(tmp_path), "adv_txt.json")))
    assert sorted(map(int, adv)) == list(range(6))
    pts = [f for _, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith(".pt")]
    assert len(pts) == 6
