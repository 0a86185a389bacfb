"""``bench.py --gpus N`` must run N ranks or fail -- never a silent 1-GPU run under an N-GPU label.

The driver may start the scaling bench either under ``torch.distributed.run`` (WORLD_SIZE set) or as a plain
``python bench.py --gpus N``; in the second case bench.py itself starts N ranks as children before any GPU call.  These
tests run on the CPU with ``--dry-run`` (launch plumbing only: gloo rendezvous on host tensors, stand-in success bits,
no measurement -- ``value`` is null); the GPU twins are tests/test_bench_multirank.py and test_rccl_single_rank.py.
Reference hook of the sharding: ``DistributedSampler(test_dataset, shuffle=False)``,
``VLMO_VQAttack/vlmo/datamodules/multitask_datamodule.py:54``.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
LAUNCHER_VARS = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VQA_BENCH_FAIL_RANK",
                 "TORCHELASTIC_RUN_ID", "GROUP_RANK", "LOCAL_WORLD_SIZE")


def _run(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    e.update(OMP_NUM_THREADS="1", **env)
    return subprocess.run([sys.executable, BENCH] + argv, env=e, capture_output=True, text=True, timeout=300)


def _json_lines(text):
    return [json.loads(ln) for ln in text.splitlines() if ln.startswith("{")]


def test_gpus_2_without_launcher_env_runs_two_ranks():
    p = _run(["--gpus", "2", "--dry-run", "--batch", "5"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout[-2000:]                     # only rank 0 reports
    rec = lines[0]
    assert rec["n_gpus"] == 2 and rec["collective"]["world"] == 2 and rec["dry_run"] is True and rec["value"] is None
    assert abs(rec["attack_success_rate"] - 4 / 10) < 1e-6       # ids 0..9 over both ranks, success iff id % 3 == 0


def test_sweep_mode_is_a_fixed_set_sharded_over_eight_ranks():
    """``bench.py --sweep N`` = BASELINE configs[3] as written: a FIXED seeded set sharded rank::world (strong scaling).
    --dry-run, 8 ranks from a plain invocation: one line from rank 0 with per-rank samples / batches / steps / gather
    time for every rank, the shards partition the set, the success rate is the one of the whole set."""
    p = _run(["--gpus", "8", "--dry-run", "--sweep", "5003", "--batch", "64"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout[-2000:]
    rec = lines[0]
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 8 and rec["dry_run"] is True and rec["value"] is None
    assert rec["config"]["n_samples"] == 5003 and rec["collective"] == {"backend": "gloo", "world": 8, "calls": 2}
    per = rec["per_rank"]
    assert per["samples"] == [626, 626, 626, 625, 625, 625, 625, 625] and sum(per["samples"]) == 5003
    assert all(len(per[k]) == 8 for k in per) and per["n_batches"] == [10] * 8
    assert sum(per["dual_loss_samples"]) in (1251, 1252, 1253)           # one in four (+ a chance hit of the decision logic)
    assert min(per["dual_loss_samples"]) >= 100                             # ... and on every rank, not on two of them
    # every sample's own schedule: 40 image steps + one probe per substitutable word but the last block's
    from vqattack_amd.attack.schedule import gradient_steps
    from vqattack_amd.attack.sweep import synthetic_questions
    ids, _, att = synthetic_questions(5003, 40, seed=0)
    assert sum(per["sample_steps"]) == sum(gradient_steps(int(n), 40) for n in att.sum(dim=1).tolist())
    assert rec["imbalance"]["sample_steps_max"] == max(per["sample_steps"])
    assert abs(rec["attack_success_rate"] - float((ids[:, 1] % 3 == 0).float().mean())) < 1e-6
    assert rec["distinct_devices"] == 1            # the dry run's ranks share the host; under RCCL it must equal n_gpus


def test_sweep_mode_single_process():
    p = _run(["--gpus", "1", "--dry-run", "--sweep", "37", "--batch", "8"])
    assert p.returncode == 0, p.stderr[-2000:]
    rec = _json_lines(p.stdout)[0]
    assert rec["n_gpus"] == 1 and rec["collective"] is None and rec["per_rank"]["samples"] == [37]
    assert rec["per_rank"]["n_batches"] == [5]


def test_distinct_device_accounting():
    """``distinct_devices``: the number of different device identities (SHA-1 halves of GPU UUID + PCI address) gathered
    over the ranks; unknown (a rank could not name its device) is reported as null and never refused; under RCCL a line
    whose ranks shared a GPU is refused, the gloo rehearsal (several ranks on one GPU on purpose) is not."""
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    a, b = (11, 12), (21, 22)
    assert bench.count_distinct([a, b, a]) == 2 and bench.count_distinct([a]) == 1
    assert bench.count_distinct([a, (0, 0)]) is None
    bench.check_distinct([a, b], 2, "nccl")
    bench.check_distinct([a, a], 2, "gloo")
    bench.check_distinct([a, (0, 0)], 2, "nccl")
    with pytest.raises(SystemExit):
        bench.check_distinct([a, a], 2, "nccl")
    recs, ids = bench.gather_rank_records([1.0, 2.0], "uuid:x|pci:0000:01:00", False, None)
    assert recs == [[1.0, 2.0]] and len(ids) == 1 and ids[0] != (0, 0)
    assert bench.gather_rank_records([0.0], None, False, None)[1] == [(0, 0)]


def test_world_size_mismatch_exits_non_zero():
    p = _run(["--gpus", "2", "--dry-run"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
             MASTER_PORT="29999")
    assert p.returncode != 0 and not _json_lines(p.stdout)
    assert "must agree" in p.stderr


def test_a_failing_rank_fails_the_run():
    p = _run(["--gpus", "2", "--dry-run"], VQA_BENCH_FAIL_RANK="1")
    assert p.returncode != 0


def test_gpus_1_stays_single_process():
    p = _run(["--gpus", "1", "--dry-run", "--batch", "3"])
    assert p.returncode == 0, p.stderr[-2000:]
    rec = _json_lines(p.stdout)[0]
    assert rec["n_gpus"] == 1 and rec["collective"] is None


def test_more_ranks_than_gpus_is_refused_for_rccl():
    """Without --dry-run the RCCL path needs one GPU per rank: on a box with fewer GPUs the launch is refused loudly."""
    import torch
    if torch.cuda.device_count() >= 8:
        return
    p = _run(["--gpus", "8", "--no-cpu-baseline"])
    assert p.returncode != 0 and "only" in p.stderr and not _json_lines(p.stdout)


def test_stopping_the_launcher_stops_its_ranks():
    """SIGTERM to ``python bench.py --gpus 2`` (a caller's timeout) reaches torch.distributed.run and both rank processes:
    nothing of the run may stay behind holding a GPU."""
    import re
    import signal
    import time
    import psutil
    e = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    e.update(OMP_NUM_THREADS="1", VQA_BENCH_DRY_SLEEP="120")
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=e, stderr=subprocess.PIPE, text=True)
    try:
        pids, deadline = set(), time.time() + 240
        while len(pids) < 2 and time.time() < deadline:          # both ranks are up and asleep
            line = p.stderr.readline()
            for pid in re.findall(r"rank \d+ pid (\d+) sleeping", line):      # two ranks may share one stderr line
                pids.add(int(pid))
        assert len(pids) == 2, "the ranks did not start"
        family = psutil.Process(p.pid).children(recursive=True)
        assert pids <= {c.pid for c in family}
        p.send_signal(signal.SIGTERM)
        assert p.wait(timeout=60) != 0
        gone, alive = psutil.wait_procs(family, timeout=30)
        assert not alive, "left behind: {}".format([(c.pid, c.name()) for c in alive])
    finally:
        if p.poll() is None:
            p.kill()
        for pid in pids:
            if psutil.pid_exists(pid):
                os.kill(pid, signal.SIGKILL)


def test_visible_gpu_count_reads_the_kfd_topology_only(tmp_path, monkeypatch):
    """The launcher's device count: KFD topology + render-node access, no GPU runtime.  A fake tree: two CPU nodes, one
    GPU of this job, one GPU whose render node is missing (another tenant's), one node the cgroup hides (unreadable)."""
    sys.path.insert(0, ROOT)
    import bench
    nodes, dev = tmp_path / "nodes", tmp_path / "dri"
    dev.mkdir()
    for i, (simd, minor) in enumerate([(0, 0), (0, 0), (1024, 128), (1024, 129), (1024, 130)]):
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text("cpu_cores_count 0\nsimd_count {}\ndrm_render_minor {}\n".format(simd, minor))
    (dev / "renderD128").write_text("")
    (dev / "renderD130").write_text("")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count(str(nodes), str(dev)) == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpu_count(str(nodes), str(dev)) == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpu_count(str(nodes), str(dev)) == 2          # a list cannot add devices
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1,-1,0")
    assert bench.visible_gpu_count(str(nodes), str(dev)) == 1          # the runtimes stop at the first invalid entry
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert bench.visible_gpu_count(str(tmp_path / "absent"), str(dev)) == 0      # no KFD: no ROCm device
    if os.geteuid() != 0:                                              # root reads through any mode
        for i in range(5):
            (nodes / str(i) / "properties").chmod(0)
        assert bench.visible_gpu_count(str(nodes), str(dev)) is None   # nothing readable: unknown, never a refusal


def test_the_launcher_never_asks_torch_for_the_device_count(monkeypatch):
    """``launch_ranks`` must not reach ``torch.cuda.device_count`` (it falls back to hipGetDeviceCount when amdsmi does
    not initialise -- then the parent has initialised HIP and may not start GPU children on these hosts)."""
    import pytest
    import torch
    sys.path.insert(0, ROOT)
    import bench

    def boom():
        raise AssertionError("the launcher asked the GPU runtime")
    monkeypatch.setattr(torch.cuda, "device_count", boom)
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 1)
    monkeypatch.delenv("VQA_DIST_BACKEND", raising=False)
    args = type("A", (), dict(dry_run=False, gpus=2))()
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(args)
    assert "only 1 GPU(s) visible" in str(e.value)


def test_recorded_traffic_is_tied_to_the_kernel_source(tmp_path, monkeypatch):
    """``roofline.traffic`` is a COPIED counter reading: it is handed out only for the recorded launch shape and only while
    the kernel's source (its .hip file + common.hpp + the C-ABI header + flags) still has the digest of the PMC pass."""
    sys.path.insert(0, ROOT)
    import bench
    from vqattack_amd.build import kernel_source_digest
    name = "vqa::stream4_kernel<vqa::StepOp, 4, 5>"
    shape = {"elements": 28311552, "op": "linf_step"}
    good = kernel_source_digest(name)
    assert good and good != kernel_source_digest("vqa::neg_cos_rows_kernel<3, true, true, 4>") and \
        kernel_source_digest("some_other_kernel") is None
    rec = dict(phase="step64", kernel=name, shape=shape, traffic_bytes=453058560, source_sha256=good)
    path = tmp_path / "pmc_traffic.json"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "PMC_TRAFFIC", "pmc_traffic.json")
    path.write_text(json.dumps([rec]))
    assert bench.recorded_traffic("stream4_kernel<vqa::StepOp", shape) == (453058560, "step64", None)
    assert bench.traffic_fields("stream4_kernel<vqa::StepOp", dict(shape, elements=1))["traffic"] is None
    path.write_text(json.dumps([dict(rec, source_sha256="0" * 64)]))
    t, _, why = bench.recorded_traffic("stream4_kernel<vqa::StepOp", shape)
    assert t is None and "stale" in why
    path.write_text(json.dumps([{k: v for k, v in rec.items() if k != "source_sha256"}]))     # a pre-digest record
    fields = bench.traffic_fields("stream4_kernel<vqa::StepOp", shape)
    assert fields["traffic"] is None and "stale" in fields["traffic_missing"]
