"""bench.py's N > 1 flow, rehearsed with two ranks that share the one GPU of the test box.

The driver launches the real thing as ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` with
one GPU per rank over RCCL; here the two ranks are forked from the pre-GPU fork server of ``tests/conftest.py`` (no exec
from a GPU-initialised process), read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from their environment exactly like
under torchrun, and use ``VQA_DIST_BACKEND=gloo`` so that both can sit on ``cuda:0`` (the collectives then run on host
tensors; the RCCL branch of bench.py is untouched).  Checked: rank 0 prints ONE JSON line for n_gpus = 2 whose value is
the aggregate over both ranks (2 x batch x steps / max-over-ranks time) and whose ASR was gathered from both ranks.
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


def _bench_rank(rank, world, port, out_path, argv):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), VQA_DIST_BACKEND="gloo")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    sys.argv = ["bench.py"] + list(argv)
    with open(out_path, "w") as out:
        os.dup2(out.fileno(), 1)                      # the JSON line goes to the file, whoever prints it
        import bench
        bench.main()
        sys.stdout.flush()


def test_bench_two_ranks_on_one_gpu(tmp_path):
    ctx = multiprocessing.get_context("forkserver")
    world, batch, steps = 2, 4, 2
    argv = ["--gpus", "2", "--steps", str(steps), "--warmup", "1", "--model", "vlmo_tiny", "--batch", str(batch),
            "--pgd-steps", "6", "--no-cpu-baseline", "--no-b256"]
    port = _free_port()
    outs = [str(tmp_path / "rank{}.out".format(r)) for r in range(world)]
    procs = [ctx.Process(target=_bench_rank, args=(r, world, port, outs[r], argv)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:
        if p.is_alive():
            p.kill()
            p.join()
            pytest.fail("a bench rank did not finish within 300 s")
        assert p.exitcode == 0
    lines = [ln for ln in open(outs[0]).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line"
    assert not [ln for ln in open(outs[1]).read().splitlines() if ln.startswith("{")], "only rank 0 reports"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == steps and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["metric"] == "adversarial_vqa_examples_per_sec" and rec["unit"] == "examples/s"
    assert rec["config"]["batch_per_gpu"] == batch
    # whole-job aggregate over the max-over-ranks time: value = world * batch * steps / (ms_per_step * steps)
    assert abs(rec["value"] - world * batch / (rec["ms_per_step"] / 1e3)) <= 0.02 * rec["value"]
    asr = rec["attack_success_rate"]
    assert asr is not None and 0.0 <= asr <= 1.0
    assert abs(asr * world * batch * steps - round(asr * world * batch * steps)) < 1e-4, "ASR is over 2 x batch x steps bits"
    assert rec["roofline"] is not None and rec["roofline"]["launches"] == 6 * steps and rec["vs_baseline"] is None


def test_sweep_two_ranks_on_one_gpu(tmp_path):
    """``bench.py --sweep 256`` (BASELINE configs[3]'s form: a fixed set, strong scaling) with two ranks rehearsing on the
    one GPU: one line, 128 samples per rank, per-rank records of both ranks, ``distinct_devices == 1`` (under RCCL the
    bench refuses a line whose ranks shared a GPU; the gloo rehearsal switch is what allows it here)."""
    ctx = multiprocessing.get_context("forkserver")
    world = 2
    argv = ["--gpus", "2", "--sweep", "256", "--warmup", "1", "--model", "vlmo_tiny", "--batch", "32", "--pgd-steps", "8"]
    port = _free_port()
    outs = [str(tmp_path / "rank{}.out".format(r)) for r in range(world)]
    procs = [ctx.Process(target=_bench_rank, args=(r, world, port, outs[r], argv)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=420)
    for p in procs:
        if p.is_alive():
            p.kill()
            p.join()
            pytest.fail("a bench rank did not finish within 420 s")
        assert p.exitcode == 0
    lines = [ln for ln in open(outs[0]).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in open(outs[1]).read().splitlines() if ln.startswith("{")]
    rec = json.loads(lines[0])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2 and rec["config"]["n_samples"] == 256
    assert rec["per_rank"]["samples"] == [128, 128] and rec["per_rank"]["n_batches"] == [4, 4]
    assert rec["distinct_devices"] == 1 and rec["collective"]["backend"] == "gloo" and rec["collective"]["calls"] == 2
    assert abs(rec["value"] - 256 / rec["seconds"]) <= 0.01 * rec["value"]
    assert rec["seconds"] >= max(rec["per_rank"]["seconds_attack"])          # max over ranks, gathers included
    asr = rec["attack_success_rate"]
    assert abs(asr * 256 - round(asr * 256)) < 1e-3, "the success rate is over the 256 gathered bits"


def _plain_invocation(out_path, argv):
    """``python bench.py --gpus 2 ...`` WITHOUT a launcher environment, from a process that has not touched the GPU (the
    fork server's child): bench.py must start the two ranks itself (torch.distributed.run as a child process)."""
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        os.environ.pop(k, None)
    os.environ.update(VQA_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")      # two ranks share the one GPU of the test box
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    sys.argv = ["bench.py"] + list(argv)
    out = open(out_path, "w")
    os.dup2(out.fileno(), 1)
    os.dup2(out.fileno(), 2)
    import bench
    bench.main()                                  # sys.exit(launcher's code)


def test_plain_gpus_2_invocation_starts_two_ranks(tmp_path):
    """The driver may run the scaling bench as a plain ``python bench.py --gpus N``: that must be an N-rank run (here
    N = 2 on one GPU, gloo for the collectives), never a silent 1-GPU run under an N-GPU label."""
    ctx = multiprocessing.get_context("forkserver")
    out = str(tmp_path / "plain.out")
    argv = ["--gpus", "2", "--steps", "1", "--warmup", "1", "--model", "vlmo_tiny", "--batch", "4", "--pgd-steps", "6",
            "--no-cpu-baseline", "--no-b256"]
    p = ctx.Process(target=_plain_invocation, args=(out, argv))
    p.start()
    p.join(timeout=420)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("the launcher did not finish within 420 s")
    text = open(out).read()
    assert p.exitcode == 0, text[-3000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-3000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["collective"]["world"] == 2 and rec["collective"]["backend"] == "gloo"
    assert rec["value"] > 0 and rec["config"]["batch_per_gpu"] == 4


def _rccl_refusal(out_path):
    """``python bench.py --gpus 2`` with the DEFAULT backend (RCCL) on a one-GPU box, from a process that has not touched
    the GPU: the launcher must refuse -- and must still be GPU-free when it does."""
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "VQA_DIST_BACKEND"):
        os.environ.pop(k, None)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    sys.argv = ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--model", "vlmo_tiny", "--batch", "2",
                "--no-cpu-baseline", "--no-b256"]
    import torch
    import bench
    report = {"count": bench.visible_gpu_count()}
    try:
        bench.main()
        report["exit"] = "returned"
    except SystemExit as e:
        report["exit"] = str(e.code)
    report["cuda_initialized"] = bool(torch.cuda.is_initialized())
    fds = []
    for f in os.listdir("/proc/self/fd"):
        try:
            fds.append(os.readlink("/proc/self/fd/" + f))
        except OSError:
            pass
    report["gpu_fds"] = [f for f in fds if f.startswith("/dev/kfd") or f.startswith("/dev/dri")]
    maps = open("/proc/self/maps").read()
    report["runtime_mapped"] = [lib for lib in ("libamdhip64", "libhsa-runtime64") if lib in maps]
    with open(out_path, "w") as fh:
        json.dump(report, fh)


def test_rccl_launcher_refuses_two_ranks_on_one_gpu_and_stays_gpu_free(tmp_path):
    """The RCCL branch of the self-launcher, executed: default backend, ``--gpus 2``, one visible GPU -> "only 1 GPU(s)
    visible", no rank started; afterwards the launcher process has no initialised CUDA context and no ``/dev/kfd`` /
    ``/dev/dri`` descriptor (``libamdhip64`` is mapped by ``import torch`` itself; what matters is that nothing was
    opened through it)."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    ctx = multiprocessing.get_context("forkserver")
    out = str(tmp_path / "refusal.json")
    p = ctx.Process(target=_rccl_refusal, args=(out,))
    p.start()
    p.join(timeout=240)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("the launcher did not return within 240 s")
    assert p.exitcode == 0
    rep = json.load(open(out))
    assert rep["count"] == 1
    assert "only 1 GPU(s) visible" in rep["exit"], rep
    assert rep["cuda_initialized"] is False and rep["gpu_fds"] == [], rep


def _single_process_bench(out_path, argv):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        os.environ.pop(k, None)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    sys.argv = ["bench.py"] + list(argv)
    out = open(out_path, "w")
    os.dup2(out.fileno(), 1)
    os.dup2(out.fileno(), 2)
    import bench
    bench.main()
    sys.stdout.flush()


def _bench_line(tmp_path, argv, name):
    ctx = multiprocessing.get_context("forkserver")
    out = str(tmp_path / name)
    p = ctx.Process(target=_single_process_bench, args=(out, argv))
    p.start()
    p.join(timeout=420)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("bench.py did not finish within 420 s")
    text = open(out).read()
    assert p.exitcode == 0, text[-3000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-3000:]
    return json.loads(lines[0])


def test_sweep_with_the_input_and_output_steps_in_the_timed_region(tmp_path):
    """``--sweep N --from-uint8``: 8-bit host images -> device resize + normalise before, ``<qid>.pt`` after every batch,
    both inside the timed region, their host-visible share reported."""
    rec = _bench_line(tmp_path, ["--gpus", "1", "--sweep", "48", "--warmup", "1", "--model", "vlmo_tiny", "--batch", "16",
                                 "--pgd-steps", "8", "--from-uint8"], "uint8.out")
    io = rec["input_pipeline"]
    assert io is not None and len(io["input_seconds"]) == 1 and io["input_seconds"][0] > 0.0
    assert 0.0 < io["share_of_seconds_attack"] < 1.0 and io["writer_seconds"][0] >= 0.0
    assert rec["scaling"] == "strong" and rec["value"] > 0 and rec["per_rank"]["samples"] == [48]


def test_emulated_shards_line_is_labelled_as_a_prediction(tmp_path):
    """``--sweep N --emulate-world 4`` on one GPU: the four rank::4 shards in sequence; the line says so
    (``strong-emulated``, ``n_gpus`` 1), carries per-shard records and the max / mean prediction -- never an N-GPU claim."""
    rec = _bench_line(tmp_path, ["--gpus", "1", "--sweep", "50", "--warmup", "1", "--model", "vlmo_tiny", "--batch", "8",
                                 "--pgd-steps", "8", "--emulate-world", "4"], "emulated.out")
    assert rec["scaling"] == "strong-emulated" and rec["n_gpus"] == 1 and rec["distinct_devices"] == 1
    em = rec["emulated"]
    assert em["world"] == 4 and rec["per_rank"]["samples"] == [13, 13, 12, 12] and rec["collective"] is None
    secs = rec["per_rank"]["seconds_attack"]
    assert abs(em["shard_seconds_max"] - max(secs)) < 2e-3 and 0.0 < em["predicted_strong_scaling_efficiency"] <= 1.0
    assert abs(em["predicted_value_at_world"] - 50 / max(secs)) <= 0.01 * em["predicted_value_at_world"]
    assert abs(rec["value"] - 50 / rec["seconds"]) <= 0.01 * rec["value"] and rec["seconds"] >= sum(secs) * 0.99
    assert "NOT a multi-GPU measurement" in em["what"]
    asr = rec["attack_success_rate"]
    assert abs(asr * 50 - round(asr * 50)) < 1e-3
