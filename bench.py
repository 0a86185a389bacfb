#!/usr/bin/env python
"""bench.py -- adversarial VQA examples/sec of the MI355X-native VQAttack PGD path.

Contract (see the task description): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver starts
it under ``python -m torch.distributed.run`` (one rank per GPU, RCCL).  A *step* is one full attack of one batch:
BASELINE.json configs[1] -- VLMO-base white box (random-init, frozen, fp32), batch 64, 40 PGD steps, 384x384 images,
40-token questions, eps 0.125 / step 0.01 / clip [-1, 1] (the reference's literals) -- followed by black-box scoring and,
for N > 1, the RCCL all-gather of attack-success bits.  Inputs are synthetic and resident in HBM before the timed
region.  Rank 0 prints ONE JSON line with the aggregate examples/sec over all ranks (weak scaling: every rank attacks
its own batch of 64), plus

  * ``roofline``: the fused sign+clamp+project kernel (``vqa_linf_step``), timed live with HIP events on the launch
    stream inside the timed region; achieved = 16 B/element x elements per launch / mean launch duration, against the
    8 TB/s HBM3E peak; ``roofline_b256`` repeats it at the batch the north-star target is stated for.
  * ``roofline_attention``: the white box's hand-written fp32 MFMA attention (``vqa_attn_fwd`` / ``vqa_attn_bwd``) at the
    attack's shape, back-to-back, against the dense fp32 matrix peak (``bound`` = "mfma").
  * ``cpu_baseline``: the CPU oracle (reference op chain, reference-style batch-1 adapters) on the host cores, on a
    bounded sample (1 image, a few of the 40 steps, extrapolated), rank 0 / N == 1 only.
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: must be in the environment BEFORE the HIP runtime initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a float4 copy reaches
STEP_BYTES_PER_ELEM = 16    # read x, grad, x0 + write x' 
FP32_MFMA_PEAK_TFS = 157.3  # dense fp32 matrix peak, v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 2); with --sweep: timed passes over the whole fixed set (default 1)")
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed warm-up steps (default 1); with --sweep: untimed warm-up BATCHES per rank (default 1)")
    ap.add_argument("--sweep", type=int, default=0, metavar="N_SAMPLES",
                    help="BASELINE configs[3] as written: a FIXED seeded set of N_SAMPLES joint attacks (mixed schedules, "
                         "every --dual-every-th sample dual-loss) sharded rank::world through attack/sweep.run_sweep -- "
                         "strong scaling: examples/s = N_SAMPLES / max-over-ranks seconds")
    ap.add_argument("--from-uint8", action="store_true",
                    help="with --sweep: the input and output steps either side of the attack INSIDE the timed region -- "
                         "seeded 8-bit 480x640 host images through the file pipeline's device path (pinned upload, "
                         "Pillow-exact resize + normalise, csrc/image.hip; ALBEF_attack/dataset/__init__.py:35-39) and "
                         "every adversarial image written as <qid>.pt (adv_attack.py:714) to a scratch directory; the "
                         "line reports their share")
    ap.add_argument("--emulate-world", type=int, default=0, metavar="W",
                    help="with --sweep on ONE GPU: run the W rank::W shards of the set one after another in this process "
                         "and report per-shard seconds / passes / steps and max/mean over shards -- a PREDICTION of the "
                         "W-GPU strong-scaling efficiency from the shard imbalance, labelled 'strong-emulated', never a "
                         "scaling measurement")
    ap.add_argument("--dual-every", type=int, default=4, help="with --sweep: one sample in n is a dual-loss sample (spread over the set by a seeded permutation)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--pgd-steps", type=int, default=40)
    ap.add_argument("--image-size", type=int, default=384)
    ap.add_argument("--model", default="vlmo_base",
                    choices=["vlmo_base", "vlmo_large", "vlmo_tiny", "albef_base", "albef_tiny"])
    ap.add_argument("--joint", type=int, default=0, metavar="WORDS",
                    help="joint image+text attack with this many substitutable words per question (configs[4]); "
                         "0 = image-only PGD (configs[1], the default metric)")
    ap.add_argument("--dual", action="store_true",
                    help="dual-loss attack (the reference's old_alg == 0 samples): every PGD iteration is a feature step + "
                         "an MLM step on the [MASK]-ed paraphrase; --pgd-steps counts white-box gradient steps")
    ap.add_argument("--dense-mlm", action="store_true",
                    help="with --dual: MLM head and cross entropy over all B x L positions (the reference's dense "
                         "closure) instead of the live label rows only -- the 'before' of profiles/r03")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=40,
                    help="PGD steps of the CPU sample (default: one full 40-step example, ~10-15 s on 16 cores)")
    ap.add_argument("--no-b256", action="store_true")
    ap.add_argument("--no-reference-style", action="store_true",
                    help="skip the batch-1 leg that drives the drop-in with the reference's own packed closure")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch plumbing only, runs without a GPU: the ranks rendezvous over gloo on the host, gather "
                         "stand-in success bits and rank 0 prints a line whose value is null (no measurement)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 1 if args.sweep else 2
    if args.warmup is None:
        args.warmup = 1
    return args


def device_identity(index=None):
    """A string that names the PHYSICAL device a rank computes on: the GPU's UUID and its PCI address, whichever the
    runtime reports (both when both).  Gathered over the ranks, the number of different strings is the number of GPUs that
    took part -- the evidence that an N-rank line was measured on N GPUs (``distinct_devices``).  None when the runtime
    reports neither (then nothing can be proven or refuted, and nothing is refused)."""
    if index is None or not torch.cuda.is_available():
        import socket
        return "host:{}".format(socket.gethostname())
    prop = torch.cuda.get_device_properties(index)
    parts = []
    uuid = getattr(prop, "uuid", None)
    if uuid is not None and str(uuid).replace("0", "").replace("-", "") != "":
        parts.append("uuid:{}".format(uuid))
    if all(hasattr(prop, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        parts.append("pci:{:04x}:{:02x}:{:02x}".format(prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id))
    return "|".join(parts) if parts else None


def gather_rank_records(record, identity, use_dist, coll_device):
    """Every rank's (float record, device identity) on every rank: two small all-gathers OUTSIDE the timed region.
    The identity travels as the two 63-bit halves of its SHA-1."""
    import hashlib
    h = hashlib.sha1(identity.encode()).digest() if identity is not None else bytes(16)     # all zero: unknown
    ident = [int.from_bytes(h[:8], "big") >> 1, int.from_bytes(h[8:16], "big") >> 1]
    if not use_dist:
        return [list(record)], [tuple(ident)]
    rec = torch.tensor(record, dtype=torch.float64, device=coll_device)
    idt = torch.tensor(ident, dtype=torch.int64, device=coll_device)
    recs = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    idts = [torch.empty_like(idt) for _ in range(dist.get_world_size())]
    dist.all_gather(recs, rec)
    dist.all_gather(idts, idt)
    return [r.cpu().tolist() for r in recs], [tuple(i.cpu().tolist()) for i in idts]


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count(sysfs="/sys/class/kfd/kfd/topology/nodes", dev="/dev/dri"):
    """GPUs this process could open, counted WITHOUT the HIP / HSA runtime (no ``torch.cuda`` call, no ``/dev/kfd``
    descriptor: ``torch.cuda.device_count()`` falls back to ``hipGetDeviceCount`` whenever amdsmi does not initialise,
    and a launcher that has initialised HIP must not start GPU children on these hosts).  Read from the KFD topology the
    kernel publishes: a node is a GPU when its ``simd_count`` is positive; it is THIS job's when its properties are
    readable (a device cgroup hides the other tenants' nodes: EPERM) and its render node ``/dev/dri/renderD<minor>`` is
    accessible -- the test the ROCr runtime itself applies when it enumerates agents.  ``*_VISIBLE_DEVICES`` lists cap
    the count.  No KFD topology at all: 0 (no ROCm device can exist).  None: the topology is there but nothing in it
    could be read -- unknown; the caller then skips its pre-check (the ranks' ``check_distinct`` still refuses a run
    whose ranks shared a device)."""
    if not os.path.isdir(sysfs):
        return 0
    count, readable = 0, 0
    for node in sorted(os.listdir(sysfs)):
        try:
            with open(os.path.join(sysfs, node, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            continue
        readable += 1
        if int(props.get("simd_count", "0")) <= 0:
            continue                                                     # a CPU node
        minor = int(props.get("drm_render_minor", "-1"))
        if minor >= 0 and os.path.isdir(dev) and not os.access(os.path.join(dev, "renderD{}".format(minor)),
                                                               os.R_OK | os.W_OK):
            continue
        count += 1
    if readable == 0:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        listed = os.environ.get(var, "").strip()
        if listed:
            entries = []
            for e in listed.split(","):                                  # the runtimes stop at the first invalid entry
                if e.strip() in ("", "-1"):
                    break
                entries.append(e)
            count = min(count, len(entries))
    return count


def launch_ranks(args):
    """``--gpus N`` (N > 1) without a launcher environment: start the N ranks as CHILD processes under
    ``torch.distributed.run`` -- before this process has made any GPU call (it never makes one: a process that has
    initialised HIP must not exec or fork GPU work on these hosts) -- let them inherit stdout / stderr (rank 0 prints
    the JSON line) and return the launcher's exit code: a failed rank fails the whole run."""
    import subprocess
    if not args.dry_run and os.environ.get("VQA_DIST_BACKEND", "nccl") == "nccl":
        have = visible_gpu_count()                     # sysfs only: this process never touches the GPU runtime
        if have is not None and have < args.gpus:
            raise SystemExit("bench.py --gpus {}: only {} GPU(s) visible; refusing to run fewer ranks under an "
                             "N-GPU label".format(args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("--gpus {} without a launcher environment: starting {} ranks: {}".format(args.gpus, args.gpus, " ".join(cmd)))
    import signal
    proc = subprocess.Popen(cmd)

    def forward(signum, _frame):            # a caller that stops this process (timeout, ^C) stops the ranks too:
        if proc.poll() is None:             # torch.distributed.run ends its workers on SIGTERM / SIGINT
            proc.send_signal(signum)
    previous = {s: signal.signal(s, forward) for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    try:
        return proc.wait()
    finally:
        for s, h in previous.items():
            signal.signal(s, h)
        if proc.poll() is None:             # leaving on an exception: no rank may outlive the launcher
            proc.terminate()
            try:
                proc.wait(timeout=30)
            except subprocess.TimeoutExpired:
                proc.kill()


def dry_run(args, world, rank):
    """The launch path without the GPU work: rendezvous (gloo, host tensors), shard, gather, one line from rank 0."""
    from vqattack_amd.attack.asr import SuccessLedger, shard_indices
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group("gloo")
    if int(os.environ.get("VQA_BENCH_FAIL_RANK", "-1")) == rank:
        raise SystemExit("rank {} fails on request (VQA_BENCH_FAIL_RANK)".format(rank))
    if os.environ.get("VQA_BENCH_DRY_SLEEP"):            # test hook: ranks that are still busy when the launcher is stopped
        log("rank {} pid {} sleeping".format(rank, os.getpid()))
        time.sleep(float(os.environ["VQA_BENCH_DRY_SLEEP"]))
    ledger = SuccessLedger(world, rank, "cpu", force_collective=dist.is_initialized())
    mine = shard_indices(world * args.batch, rank, world)
    ledger.record(torch.tensor([i % 3 == 0 for i in mine]), sample_ids=mine)
    asr = ledger.all_gather_rate(world * args.batch)
    if rank == 0:
        print(json.dumps({"metric": "adversarial_vqa_examples_per_sec", "value": None, "unit": "examples/s",
                          "n_gpus": dist.get_world_size() if dist.is_initialized() else 1, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "attack_success_rate": asr,
                          "collective": ({"backend": dist.get_backend(), "world": dist.get_world_size(),
                                          "calls": ledger.collectives} if dist.is_initialized() else None)}),
              flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


class _DryAttack:
    """--dry-run stand-in of BatchedVQAttack for the sweep plumbing (shard -> batch -> ledger -> gathers) on the host:
    no white box; a sample 'costs' its schedule's gradient steps, the first body token is marked as substituted."""

    class cfg:
        budget = 40

    def attack_mixed(self, images, text_ids, text_masks, attackable, **_kw):
        from vqattack_amd.attack.runner import BatchResult
        from vqattack_amd.attack.schedule import gradient_steps
        per = [gradient_steps(int(n), self.cfg.budget) for n in attackable.sum(dim=1).tolist()]
        adv = text_ids.clone()
        adv[:, 1] = -adv[:, 1]
        return BatchResult(adv_images=images, adv_text_ids=adv, gradient_steps=sum(per), sample_steps=sum(per),
                           global_steps=max(per))


class _DryBlack:
    """The stand-in victim changes its answer iff the ORIGINAL first body token is divisible by 3."""

    def vqa_answer(self, images, text_ids, text_masks):
        t = text_ids[:, 1]
        return ((t < 0) & ((-t) % 3 == 0)).long()


def sweep_line(args, world, res_list, dt, records, idents, backend, flavor, text_len, image_size, dry=False,
               gemms_tuned=False):
    """The ONE line of a --sweep run (rank 0).  ``records``: per rank [attack seconds, samples, batches, sample gradient
    steps, white-box passes, gather seconds, dual-loss samples]."""
    res = res_list[-1]
    n = args.sweep
    col = lambda j: [r[j] for r in records]                                  # noqa: E731
    seconds, n_local, n_batches, s_steps, g_steps, gather, n_dual = (col(j) for j in range(7))
    has_io = all(len(r) >= 10 for r in records)
    io_in, io_blocked, io_writer = (col(j) for j in range(7, 10)) if has_io else ([0.0], [0.0], [0.0])
    distinct = count_distinct(idents)
    return {
        "metric": "adversarial_vqa_examples_per_sec", "value": None if dry else round(n * args.steps / dt, 4),
        "unit": "examples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": None if dry else round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "{} VQAttack sweep (BASELINE configs[3]{}): a FIXED seeded set of {} (image, question) pairs "
                               "sharded rank::world over {} rank(s), joint image+text attack with each sample's own block "
                               "schedule (40 image steps + one probe per substitutable word, 4..12 words), one sample in {} "
                               "dual-loss, mixed batches of <= {} per rank sorted by schedule length, {}x{} images, "
                               "black-box scoring, ONE all-gather of success bits + ONE of the adversarial text; one step "
                               "= one pass over the whole set".format(
                                   args.model, "" if (args.model == "vlmo_base" and n >= 5000 and image_size == 384
                                                      and args.pgd_steps == 40) else " shape, reduced",
                                   n, world, args.dual_every, args.batch, image_size, image_size),
                   "n_samples": n, "batch_per_gpu": args.batch, "pgd_steps": args.pgd_steps, "image_size": image_size,
                   "text_len": text_len, "dual_every": args.dual_every,
                   "sharding": "rank::world (DistributedSampler(shuffle=False), multitask_datamodule.py:54)"},
        "dry_run": dry, "attack_success_rate": res["asr"], "distinct_devices": distinct,
        "per_rank": {"seconds_attack": [round(v, 3) for v in seconds], "samples": [int(v) for v in n_local],
                     "dual_loss_samples": [int(v) for v in n_dual],
                     "n_batches": [int(v) for v in n_batches], "sample_steps": [int(v) for v in s_steps],
                     "white_box_passes": [int(v) for v in g_steps],
                     "gather_seconds": [round(v, 4) for v in gather]},
        "imbalance": {"seconds_attack_min": round(min(seconds), 3), "seconds_attack_max": round(max(seconds), 3),
                      "sample_steps_min": int(min(s_steps)), "sample_steps_max": int(max(s_steps)),
                      "white_box_passes_min": int(min(g_steps)), "white_box_passes_max": int(max(g_steps)),
                      "note": "a rank's gather_seconds includes its wait for the slowest rank; the collective's own "
                              "latency is the minimum over ranks"},
        "all_gather_latency_ms": round(min(gather) * 1e3, 3),
        # --from-uint8: 8-bit host images -> device resize + normalise before, <qid>.pt files after every batch, both inside
        # seconds_attack.  input_seconds = host time of producing the image batches (decode wait + upload + kernel
        # launches), of which input_blocked_seconds waited for pixels; writer_seconds = the wait for the asynchronous
        # writer after the last batch (its copies and torch.save calls otherwise overlap the next batch's attack)
        "input_pipeline": ({"source": "seeded uint8 (480, 640, 3) host arrays, one per sample", "resize": "Pillow-exact "
                            "bicubic + ToTensor + Normalize(0.5, 0.5) on the device (csrc/image.hip)",
                            "writer": "<qid>.pt (1, 3, H, W) fp32 per sample, scratch directory",
                            "input_seconds": [round(v, 3) for v in io_in],
                            "input_blocked_seconds": [round(v, 3) for v in io_blocked],
                            "writer_seconds": [round(v, 3) for v in io_writer],
                            "share_of_seconds_attack": round(max((a + w) / s for a, w, s in zip(io_in, io_writer, seconds)
                                                                 if s > 0), 4)}
                           if getattr(args, "from_uint8", False) and has_io else None),
        "tuned_gemms": gemms_tuned,
        "collective": ({"backend": backend, "world": world, "calls": res["collectives"]} if backend else None),
    }


def count_distinct(idents):
    """Number of different devices among the ranks' identities; None when a rank could not name its device."""
    return None if any(i == (0, 0) for i in idents) else len(set(idents))


def check_distinct(idents, world, backend):
    n = count_distinct(idents)
    if backend == "nccl" and n is not None and n != world:
        raise SystemExit("bench.py: {} ranks ran on {} distinct GPU(s): refusing to print an N-GPU line".format(world, n))


def dry_run_sweep(args, world, rank):
    """--sweep --dry-run: the strong-scaling plumbing on the host (gloo): fixed set, rank::world shards, mixed batches,
    success bits and adversarial text gathered, per-rank records gathered, one line from rank 0 (value null)."""
    from vqattack_amd.attack.sweep import run_sweep
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        dist.init_process_group("gloo")
    torch.set_num_threads(1)
    res = run_sweep("vlmo", None, _DryBlack(), None, n_samples=args.sweep, batch=args.batch, image_size=8, text_len=40,
                    device="cpu", rank=rank, world=world, log_every=0, dual_every=args.dual_every, mixed=True,
                    attack=_DryAttack(), force_collective=use_dist)
    record = [res["seconds"], res["n_local"], res["n_batches"], res["gradient_steps"], res["global_steps"],
              res["gather_seconds"], res["n_dual_local"]]
    records, idents = gather_rank_records(record, device_identity(None), use_dist, torch.device("cpu"))
    if rank == 0:
        print(json.dumps(sweep_line(args, world, [res], 1.0, records, idents, "gloo" if use_dist else None, "vlmo", 40, 8,
                                    dry=True)), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def synthetic_questions(batch, length, seed, device, min_words=4, max_words=12):
    """[CLS] + n body ids U{1000..30521} + [SEP] + padding, n ~ U{min_words..max_words} per question, seeded
    (n ~ U{4..12}).  Returns (ids, masks, words per question)."""
    g = torch.Generator().manual_seed(seed)
    hi = max(1, min(max_words, length - 2))
    lo = max(1, min(min_words, hi))
    n = torch.randint(lo, hi + 1, (batch,), generator=g)
    body = torch.randint(1000, 30522, (batch, hi), generator=g)
    ids = torch.zeros(batch, length, dtype=torch.long)
    ids[:, 0] = 101
    for s in range(batch):
        k = int(n[s])
        ids[s, 1:1 + k] = body[s, :k]
        ids[s, 1 + k] = 102
    return ids.to(device), (ids != 0).long().to(device), n.tolist()


def make_config(args):
    from vqattack_amd.whitebox import albef, vlmo
    if args.model == "vlmo_tiny":
        return vlmo.vlmo_tiny()
    if args.model == "albef_tiny":
        return albef.albef_tiny()
    if args.model == "albef_base":
        return albef.albef_base(image_size=args.image_size)
    return getattr(vlmo, args.model)(image_size=args.image_size)


def build_models(args, cfg, device):
    """(flavor, white box, black box, adapters, text length)."""
    if args.model.startswith("albef"):
        from vqattack_amd.whitebox.albef import AlbefAttackAdapters, FrozenAlbef
        white = FrozenAlbef(cfg, seed=0).to(device)
        black = FrozenAlbef.finetuned_from(white, seed=1).to(device)
        return "albef", white, black, AlbefAttackAdapters(white), (8 if args.model == "albef_tiny" else 40)
    from vqattack_amd.whitebox.vlmo import FrozenVlmo, VlmoAttackAdapters
    white = FrozenVlmo(cfg, seed=0).to(device)
    black = FrozenVlmo.finetuned_from(white, seed=1).to(device)   # VQA model = pre-trained trunk + drift + answer head
    return "vlmo", white, black, VlmoAttackAdapters(white), cfg.max_text_len


# HBM traffic per launch is NOT measured by this script (PMC counters need their own rocprofv3 passes): the tracked JSON
# below holds, per phase of tools/pmc_step.py, the launch SHAPE it ran, FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE per
# launch and kernel, and the sha256 of the kernel's source (tools/pmc_run.sh on the shipped build).  A figure is copied
# into the line -- as `traffic` with `traffic_kind: "recorded"` -- only when the run's launch has exactly the recorded
# shape AND the kernel's source in this tree still has the recorded digest; after a kernel edit without a new PMC pass
# the field is null and says why.
PMC_TRAFFIC = next((p for p in ("profiles/r06/pmc_traffic.json", "profiles/r05/pmc_traffic.json")
                    if os.path.exists(os.path.join(ROOT, p))), "profiles/r06/pmc_traffic.json")


def recorded_traffic(kernel, shape):
    """(HBM bytes per launch, phase, why-not) of the record whose kernel name contains `kernel`, whose shape dict equals
    `shape` and whose source digest is this tree's; (None, None, reason) otherwise."""
    from vqattack_amd.build import kernel_source_digest
    try:
        records = json.load(open(os.path.join(ROOT, PMC_TRAFFIC)))
    except (OSError, ValueError):
        return None, None, "no record file"
    why = "no record of this kernel and launch shape"
    for rec in records:
        if kernel in rec.get("kernel", "") and rec.get("shape") == shape:
            if rec.get("source_sha256") is None or rec["source_sha256"] != kernel_source_digest(rec["kernel"]):
                why = "stale: the kernel's source changed since the PMC pass (phase {})".format(rec.get("phase"))
                continue
            return int(rec["traffic_bytes"]), rec.get("phase"), None
    return None, None, why


def traffic_fields(kernel, shape):
    t, phase, why = recorded_traffic(kernel, shape) if shape else (None, None, "no shape")
    if t is None:
        return dict(traffic=None, traffic_source=PMC_TRAFFIC, traffic_shape=shape, traffic_missing=why)
    return dict(traffic=t, traffic_kind="recorded", traffic_shape=shape,
                traffic_source="{} (phase {}: the same launch shape, the same kernel source)".format(PMC_TRAFFIC, phase))


class KernelTimer:
    """HIP-event timing of every launch of two hand-written kernels, on the stream they are launched on:
    ``vqa_linf_step`` (the fused sign+clamp+project step) and ``vqa_neg_cos_rows`` (fused cosine loss + gradient)."""

    def __init__(self):
        self.step_events, self.step_numel = [], 0
        self.loss_events, self.loss_rows, self.loss_shape = [], [], None
        self._live = {}      # id(weight plane) -> (plane kept alive, device-side count of live rows): no host read here

    def install(self):
        from vqattack_amd import ops
        self._step, self._loss = ops.linf_step, ops.neg_cos_rows_multi
        timer = self

        def timed_step(x, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                       # torch's current stream == the launch stream (ops.stream_for)
            out = timer._step(x, *a, **kw)
            e1.record()
            timer.step_events.append((e0, e1))
            timer.step_numel = x.numel()
            return out

        def timed_loss(a_list, b_list, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = timer._loss(a_list, b_list, *a, **kw)
            e1.record()
            timer.loss_events.append((e0, e1))
            w = kw.get("row_weight")
            a_ = a_list[0]
            rows = a_.numel() // a_.shape[-1]
            live, per = None, 1
            if w is not None:                 # live (weight != 0) rows, counted on the device, read after the timed region
                if id(w) not in timer._live:  # the plane is kept alive, so its id cannot be recycled for another mask
                    timer._live[id(w)] = (w, (w != 0).sum())
                live, per = timer._live[id(w)][1], rows // w.numel()
            timer.loss_rows.append((live if live is not None else rows, per,
                                    (12 if out is not None else 8) * a_.shape[-1] * len(a_list)))
            timer.loss_shape = (tuple(a_.shape), len(a_list), w is not None)
            return out
        ops.linf_step, ops.neg_cos_rows_multi = timed_step, timed_loss

    def remove(self):
        from vqattack_amd import ops
        ops.linf_step, ops.neg_cos_rows_multi = self._step, self._loss

    def _loss_shape_record(self):
        """Shape of the (single kind of) loss launch of this run, as tools/pmc_step.py records it; None when the run
        launched more than one shape (ALBEF: text and image modality)."""
        if self.loss_shape is None or len({(int(r), p, b) for r, p, b in self.loss_rows}) != 1:
            return None
        shape, maps, weighted = self.loss_shape
        rows, per, _ = self.loss_rows[0]
        return dict(op="neg_cos_rows_multi", maps=maps, batch=shape[0], tokens=shape[1], dim=shape[2],
                    live_rows=int(rows) * per if weighted else int(rows))

    @staticmethod
    def _stats(events):
        ms = [a.elapsed_time(b) for a, b in events]
        return (sum(ms) / len(ms), min(ms), len(ms)) if ms else (None, None, 0)

    def summary(self):
        torch.cuda.synchronize()
        mean_ms, min_ms, n = self._stats(self.step_events)
        if not n:
            return None, None
        nbytes = STEP_BYTES_PER_ELEM * self.step_numel
        gbs = nbytes / mean_ms / 1e6
        step = dict(kernel="vqa_linf_step (stream4_kernel<StepOp>)", bound="hbm", achieved=round(gbs, 1),
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), launches=n,
                    mean_launch_us=round(mean_ms * 1e3, 2), min_launch_us=round(min_ms * 1e3, 2),
                    algorithmic_bytes_per_launch=nbytes,
                    timing="hip events on the launch stream, per launch; HBM traffic is not measured in this run "
                           "(separate rocprofv3 --pmc passes, summary in traffic_source)",
                    **traffic_fields("StepOp", dict(op="linf_step", elements=self.step_numel)))
        loss = None
        mean_ms, min_ms, n = self._stats(self.loss_events)
        if n:
            # launches differ in size (text / image rows): total bytes over total time; live-row counts read only now
            total_bytes = sum(int(rows) * per * row_bytes for rows, per, row_bytes in self.loss_rows)
            gbs = total_bytes / (mean_ms * n) / 1e6
            loss = dict(kernel="vqa_neg_cos_rows_multi (neg_cos_rows_full_kernel: every per-layer map of a modality in "
                               "one launch, loss folded in the launch)", bound="hbm", achieved=round(gbs, 1),
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), launches=n,
                        mean_launch_us=round(mean_ms * 1e3, 2), min_launch_us=round(min_ms * 1e3, 2),
                        algorithmic_bytes_per_launch=round(total_bytes / n),
                        note="12*D bytes per live row (read a, b; write grad) -- the padded text rows (ragged questions "
                             "inside the trimmed layout, < 1 % of the rows) are not counted: they need only their zero "
                             "gradient row, although the branch-free kernel reads them as well",
                        **traffic_fields("neg_cos_rows", self._loss_shape_record()))
        return step, loss


def warm_up(fn, seconds=0.08, chunk=8):
    """Launch ``fn`` back-to-back for at least ``seconds`` of wall clock before a timed group.  After a host-side gap of
    a few milliseconds the card idles down and the first 15-30 ms of launches run at lower clocks
    (profiles/r05/attn_harness_ab.jsonl: the SAME forward launch takes 0.98 ms right after 2 s of idle, 0.79 ms sixteen
    launches later and 0.705 ms in steady state; round 4's 2 warm-up launches + 10 timed ones sat inside that ramp and
    read 0.87-0.88 ms where the kernel trace of the attack shows 0.68 ms)."""
    t0 = time.perf_counter()
    while True:
        for _ in range(chunk):
            fn()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 >= seconds:
            return


def step_kernel_microbench(batch, image_size, reps=40):
    """Back-to-back launches of the fused step at `batch` (north-star target batch 256), one event pair around all."""
    from vqattack_amd import ops
    shape = (batch, 3, image_size, image_size)
    gen = torch.Generator(device="cuda").manual_seed(1)
    x0 = torch.empty(shape, device="cuda").uniform_(-1, 1, generator=gen)
    x = torch.clamp(x0 + torch.empty(shape, device="cuda").uniform_(-0.125, 0.125, generator=gen), -1, 1)
    g = torch.randn(shape, device="cuda", generator=gen)
    bufs = [x, torch.empty_like(x)]
    turn = [0]

    def one():
        i = turn[0]
        turn[0] += 1
        ops.linf_step(bufs[i & 1], g, x0, 0.01, 0.125, -1, 1, out=bufs[1 - (i & 1)])
    warm_up(one)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        one()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = STEP_BYTES_PER_ELEM * x.numel()
    gbs = nbytes / ms / 1e6
    return dict(kernel="vqa_linf_step", batch=batch, bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS,
                unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), mean_launch_us=round(ms * 1e3, 2),
                algorithmic_bytes_per_launch=nbytes,
                timing="{} back-to-back launches between two hip events, after >= 80 ms of the same launches "
                       "(clock ramp after host-side idle)".format(reps),
                **traffic_fields("StepOp", dict(op="linf_step", elements=x.numel())))


GUIDE_COPY_GBS = 6290.0     # float4 copy measured by MI355X_MICROARCH.md on this chip (the guide's "achievable" figure)


def stream_probe(nbytes=452984832, reps=12):
    """The platform's streaming rate on THIS box, from the best plain stream of tools/stream_probe.hip (built as
    tools/libstream_probe.so): a 3-read-1-write stream and a copy at the step kernel's batch-256 footprint (453 MB per
    buffer, 1.8 GB per launch -- beyond the 256 MB Infinity Cache), best of a few launch shapes each.  Reported next to
    the 8 TB/s spec the fractions are priced against and the guide's measured float4 copy (6.29 TB/s)."""
    import ctypes
    path = os.path.join(ROOT, "tools", "libstream_probe.so")
    if not os.path.exists(path):      # built by __graft_entry__.build() only: this process has initialised the GPU and
        raise RuntimeError("{} is absent (run __graft_entry__.build())".format(os.path.relpath(path, ROOT)))   # must not spawn a compiler
    lib = ctypes.CDLL(path)
    fn = lib.vqa_probe_stream
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_int, ctypes.c_size_t] + [ctypes.c_int] * 5
    torch.cuda.synchronize()
    shapes = [(8, 4, 0), (8, 4, 1), (8, 4, 3), (12, 4, 1), (12, 4, 3), (16, 8, 3), (32, 4, 0), (16, 4, 1)]   # (WG per CU, unroll, nt)
    out = {}
    for name, op, streams in (("step_like_3r1w", 3, 4), ("copy_1r1w", 2, 2)):
        best = max(((fn(op, nbytes, per_cu, unroll, nt, 0, reps), (per_cu, unroll, nt)) for per_cu, unroll, nt in shapes))
        if best[0] < 0:
            raise RuntimeError("stream probe failed with code {}".format(best[0]))
        out[name] = dict(achieved=round(best[0], 1), unit="GB/s", frac_of_peak=round(best[0] / HBM_PEAK_GBS, 4),
                         bytes_per_launch=nbytes * streams, workgroups_per_cu=best[1][0], unroll=best[1][1],
                         nt_mask=best[1][2])
    out["kernel"] = "tools/stream_probe.hip stream_kernel (plain float4 streams, best of {} launch shapes)".format(
        len(shapes))
    out["guide_float4_copy"] = dict(achieved=GUIDE_COPY_GBS, unit="GB/s", source="MI355X_MICROARCH.md")
    return out


def attention_traffic(batch, heads, seq, with_bias):
    """Recorded HBM bytes of one forward + one backward call (the record of tools/pmc_step.py's `attn` phase with exactly
    this shape), per kernel."""
    shape = dict(op="attention", batch=batch, heads=heads, seq=seq, head_dim=64, bias=bool(with_bias))
    parts = {k: recorded_traffic(k, shape)[0] for k in ("attn_fwd_kernel", "attn_delta_kernel", "attn_bwd_dkv_kernel",
                                                        "attn_bwd_dq_staged_kernel")}
    if any(v is None for v in parts.values()):
        return dict(traffic=None, traffic_source=PMC_TRAFFIC, traffic_shape=shape)
    return dict(traffic=sum(parts.values()), traffic_kind="recorded", traffic_per_kernel=parts, traffic_shape=shape,
                traffic_source="{} (phase attn: the same four launches)".format(PMC_TRAFFIC))


def attention_microbench(batch, heads, seq, with_bias, reps=40):
    """The white box's fp32 MFMA attention (csrc/attn.hip) at the attack's shape: `reps` back-to-back forward launches
    (saving the scores, as a differentiated forward does) and `reps` backward calls (delta pre-pass, dK / dV kernel that
    starts from the saved scores and stores dS^T, dQ-from-dS^T kernel staged through LDS), one event pair around each group.  Flops are the algorithm's (2 S^2 d per matrix product and (batch, head): 2 products forward, 5
    backward -- the usual flash-attention accounting), against the dense fp32 matrix peak of v_mfma_f32_32x32x2_f32."""
    from vqattack_amd import attention
    gen = torch.Generator(device="cuda").manual_seed(2)
    qkv = torch.randn(batch, seq, 3, heads, 64, device="cuda", generator=gen)
    bias, bstr = None, None
    if with_bias:                                  # one (1, H, S, S) slab shared over the batch, rows padded to 32
        store = torch.zeros(1, heads, seq, (seq + 31) // 32 * 32, device="cuda")
        store[..., :seq] = torch.randn(1, heads, seq, seq, device="cuda", generator=gen) * 0.02
        bias = store[..., :seq].expand(batch, -1, -1, -1)
        bstr = (bias.stride(0), bias.stride(1), bias.stride(2))
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    # what the attack's autograd path runs: a forward that saves its scores, a backward that starts from them
    o, lse, scores = attention._forward(q, k, v, bias, bstr, 0.125, save_scores=True)
    go = torch.randn(o.shape, device="cuda", generator=gen)
    dqkv = torch.empty_like(qkv)

    def fwd():
        attention._forward(q, k, v, bias, bstr, 0.125, save_scores=True)

    def bwd():
        attention._backward(q, k, v, bias, bstr, o, lse, go, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], 0.125,
                            scores=scores)

    def timed(fn):
        warm_up(fn)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    product = 2.0 * batch * heads * seq * seq * 64
    ms_f, ms_b = timed(fwd), timed(bwd)
    tf_f, tf_b = 2 * product / ms_f / 1e9, 4 * product / ms_b / 1e9
    tf = 6 * product / (ms_f + ms_b) / 1e9
    return dict(kernel="vqa_attn_fwd + vqa_attn_bwd (attn_fwd_kernel; attn_delta_kernel, attn_bwd_dkv_kernel, "
                       "attn_bwd_dq_staged_kernel)",
                bound="mfma", achieved=round(tf, 1), peak=FP32_MFMA_PEAK_TFS, unit="TFLOP/s",
                frac=round(tf / FP32_MFMA_PEAK_TFS, 4), **attention_traffic(batch, heads, seq, with_bias),
                shape=dict(batch=batch, heads=heads, seq=seq, head_dim=64, bias=bool(with_bias)),
                forward=dict(ms=round(ms_f, 3), achieved=round(tf_f, 1), frac=round(tf_f / FP32_MFMA_PEAK_TFS, 4)),
                backward=dict(ms=round(ms_b, 3), achieved=round(tf_b, 1), frac=round(tf_b / FP32_MFMA_PEAK_TFS, 4)),
                flops_per_call=6 * product,
                flash_accounting=dict(products=7, achieved=round(7 * product / (ms_f + ms_b) / 1e9, 1),
                                      frac=round(7 * product / (ms_f + ms_b) / 1e9 / FP32_MFMA_PEAK_TFS, 4)),
                note="saved scores and the dS^T workspace each cross HBM once each way ({:.2f} GB per direction)".format(
                    2 * 4 * batch * heads * ((seq + 127) // 128 * 128) * ((seq + 31) // 32 * 32) / 1e9),
                timing="{} back-to-back launches per direction between two hip events, each group after >= 80 ms of the "
                       "same launches (round 4 timed 10 launches after 2 warm-up launches, inside the card's clock ramp "
                       "after host-side idle: profiles/r05/attn_harness_ab.jsonl); per-kernel durations inside the attack: "
                       "profiles/r05/bench_default_summary.txt".format(reps))


def reference_style_leg(flavor, white, cfg, text_len, pgd_steps, reps=2):
    """What a user of the reference gets from switching only the ``cleverhans`` import: the
    reference's own call -- batch 1, its ``pgd_attack`` member as ``model_fn`` (torch.stack / cat-packed plain tensors,
    ``[0]`` indexing: whitebox/reference_style.py after vlmo_module.py:1387-1446 / adv_attack.py:119-126),
    ``pgd.projected_gradient_descent(model_fn, x, 0.125, 0.01, 40, np.inf, -1, 1, y=..., time=0, ori_x=..., ls=1)``
    (adv_attack.py:607-610) -- through the drop-in operator, in ms per PGD iteration, next to the bundled batched
    adapters (``LayerFeatures``, nothing packed) at the same batch 1."""
    import numpy as np
    from vqattack_amd import dropin
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox import reference_style
    dev = next(white.parameters()).device
    ids, masks, _ = synthetic_questions(1, text_len, seed=7, device=dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    image = torch.empty(1, 3, cfg.image_size, cfg.image_size, device=dev).uniform_(-1, 1, generator=gen)
    pgd = dropin.load(flavor).projected_gradient_descent.projected_gradient_descent
    batch = dict(text_ids=ids, text_masks=masks)
    if flavor == "vlmo":
        me = reference_style.VlmoReferenceClosures(white, batch)
        y = me.Gen_ori_feats(image)
    else:
        me = reference_style.AlbefReferenceClosures(white, batch)
        img_feats, txt_feats = me.Gen_ori_feats(image)
        y = [txt_feats, img_feats, None, None, None]

    replay = bool(getattr(me, "capturable", False))     # closures without host reads: iterations 1.. replay ONE hipGraph

    def reference_call(graph=replay):
        with torch.enable_grad():
            return pgd(me.pgd_attack, image, 0.125, 0.01, pgd_steps, np.inf, -1, 1, y=list(y), time=0, ori_x=image, ls=1,
                       **(dict(graph=True) if graph else {}))

    if flavor == "vlmo":
        from vqattack_amd.whitebox.vlmo import VlmoAttackAdapters as Adapters
    else:
        from vqattack_amd.whitebox.albef import AlbefAttackAdapters as Adapters
    bundled = BatchedVQAttack(Adapters(white), flavor, white.embedding_tables(),
                              AttackConfig(budget=pgd_steps, random_start=True, sanity_checks=True))
    bundled_graph = BatchedVQAttack(Adapters(white), flavor, white.embedding_tables(),
                                    AttackConfig(budget=pgd_steps, random_start=True, sanity_checks=True, use_graph=True))
    words = torch.zeros_like(ids, dtype=torch.bool)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps / pgd_steps * 1e3
    ms_ref = timed(reference_call)
    ms_eager = timed(lambda: reference_call(False)) if replay else ms_ref
    ms_bundled = timed(lambda: bundled.attack_batch(image, ids, masks, words))
    try:
        ms_bundled_graph = round(timed(lambda: bundled_graph.attack_batch(image, ids, masks, words)), 3)
    except Exception as exc:             # noqa: BLE001  (an auxiliary figure must not lose the leg)
        ms_bundled_graph = "{}: {}".format(type(exc).__name__, str(exc)[:120])
    return dict(what="the reference's own call at its own batch size 1: its pgd_attack member (packed plain tensors, [0] "
                     "indexing) as model_fn through the drop-in projected_gradient_descent, {} steps, time=0, "
                     "sanity_checks on".format(pgd_steps),
                ms_per_pgd_iteration=round(ms_ref, 3), examples_per_sec=round(1e3 / (ms_ref * pgd_steps), 3),
                graph_replay=replay, eager_ms_per_pgd_iteration=round(ms_eager, 3),
                how="whitebox/reference_style.py keeps what depends on the text batch only (real-token index, attention "
                    "masks) per text batch instead of per forward, so the closure has no host read and the drop-in replays "
                    "iterations 1.. from one hipGraph (graph=True, an extension kwarg); at batch 1 the attention kernels cut "
                    "their tile loops into parts (attention.loop_split) so that 12 heads fill the chip",
                bundled_adapters_ms_per_pgd_iteration=round(ms_bundled, 3),
                bundled_adapters_graph_ms_per_pgd_iteration=ms_bundled_graph,
                bundled_adapters_examples_per_sec=round(1e3 / (ms_bundled * pgd_steps), 3),
                loss_launches_per_iteration=2, note="wall clock incl. the one host read per PGD call; the two (output, "
                "target) pairs of the packed form have different shapes (per-layer [CLS] rows / all token rows; text / "
                "image maps), so they are two loss launches, not one pointer-table launch")


def baseline_config(args):
    """Which entry of BASELINE.json's ``configs`` the run is (the default run is configs[1])."""
    full = args.pgd_steps == 40 and args.image_size == 384
    if args.dual:
        return "not a BASELINE config: dual-loss variant"
    if full and args.model == "vlmo_base" and not args.joint and args.batch == 64:
        return "BASELINE configs[1]" if int(os.environ.get("WORLD_SIZE", "1")) == 1 else "BASELINE configs[3] shape"
    if full and args.model == "albef_base" and not args.joint and args.batch == 256:
        return "BASELINE configs[2]"
    if full and args.model == "vlmo_large" and args.joint and args.batch == 128:
        return "BASELINE configs[4] workload"
    return "not a BASELINE config"


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (the GPU box gives a
    one-GPU job a 16-core share of a much larger host; oversubscribing torch's thread pool stalls it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("VQA_CPU_BASELINE_CORES", "16"))))


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline(args, cfg):
    """The reference's CPU path (oracle restatement + reference-style batch-1 packing) on the host cores."""
    import numpy as np
    from oracle import cleverhans_cpu as oracle
    from oracle.adapters_ref import VlmoRefAdapters
    from vqattack_amd.whitebox.vlmo import FrozenVlmo
    cores = usable_cores()
    torch.set_num_threads(cores)
    log("cpu_baseline: {} threads, {} of {} PGD steps on 1 image".format(cores, args.cpu_baseline_steps,
                                                                         args.pgd_steps))
    model = FrozenVlmo(cfg, seed=0)
    ids, masks, _ = synthetic_questions(1, cfg.max_text_len, seed=0, device="cpu")
    g = torch.Generator().manual_seed(0)
    x0 = torch.empty(1, 3, cfg.image_size, cfg.image_size).uniform_(-1, 1, generator=g)
    ad = VlmoRefAdapters(model, ids, masks)
    y = ad.gen_ori_feats(x0)
    steps = max(1, min(args.cpu_baseline_steps, args.pgd_steps))
    t0 = time.perf_counter()
    with torch.enable_grad():
        oracle.projected_gradient_descent(ad.pgd_attack, x0, 0.125, 0.01, steps, np.inf, clip_min=-1, clip_max=1,
                                          y=y, ori_x=x0, time=0, ls=1, flavor="vlmo")
    dt = time.perf_counter() - t0
    per_example = dt * args.pgd_steps / steps
    return dict(value=round(1.0 / per_example, 5), unit="examples/s", cores=cores, kind="port",
                sample="1 image x {} of {} PGD steps ({:.1f} s of CPU work{}); oracle/cleverhans_cpu.py + "
                       "reference-style batch-1 adapter on torch CPU fp32".format(
                           steps, args.pgd_steps, dt,
                           "" if steps == args.pgd_steps else ", extrapolated to {} steps".format(args.pgd_steps)))


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:                 # plain `python bench.py --gpus N`: never a silent 1-GPU run under an N-GPU label
            sys.exit(launch_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py --gpus {} was launched with WORLD_SIZE={}: the launcher's --nproc-per-node and --gpus "
                         "must agree".format(args.gpus, os.environ["WORLD_SIZE"]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.dry_run:
        return dry_run_sweep(args, world, rank) if args.sweep else dry_run(args, world, rank)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)   # any torchrun launch, also N = 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the attack path has no CPU fallback")
    # rehearsal mode: VQA_DIST_BACKEND=gloo lets several ranks share one GPU (collectives then run on host tensors);
    # the driver's multi-GPU runs use the default, RCCL ("nccl") with one GPU per rank
    backend = os.environ.get("VQA_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    if use_dist:
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a rank that never arrives must end the run with an error line well inside the caller's limit, not hang it
        limit = datetime.timedelta(seconds=float(os.environ.get("VQA_DIST_TIMEOUT", "240")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    identity = device_identity(dev_index)
    # the devices are known now: a run whose ranks share a GPU under RCCL is refused BEFORE the warm-up and the timed
    # region burn the lease (the gather after the run only collects the per-rank records)
    _, idents = gather_rank_records([0.0], identity, use_dist, coll_device)
    check_distinct(idents, world, backend if use_dist else None)

    from vqattack_amd.attack.asr import SuccessLedger
    from vqattack_amd.attack.runner import AttackConfig, BatchedVQAttack
    from vqattack_amd.whitebox import tuned_gemms

    # recorded hipBLASLt solutions for the white boxes' fp32 GEMMs (read-only TunableOp; library defaults when the tracked
    # file is absent or was recorded for another software stack) -- VQA_TUNED_GEMMS=off disables it
    gemms_tuned = tuned_gemms.enable()
    cfg = make_config(args)
    flavor, white, black, adapters, text_len = build_models(args, cfg, device)
    attack = BatchedVQAttack(adapters, flavor, white.embedding_tables(),
                             AttackConfig(budget=args.pgd_steps, random_start=True, sanity_checks=True,   # the reference's
                                          live_mlm_rows=not args.dense_mlm))   # call sites use the default True (adv_attack.py:633-636)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.sweep:
        # ---- BASELINE configs[3] as written: a fixed set, sharded; strong scaling
        from vqattack_amd.attack.sweep import run_sweep
        sweep_attack = BatchedVQAttack(adapters, flavor, white.embedding_tables(),
                                       AttackConfig(budget=args.pgd_steps, random_start=True))

        last = [time.perf_counter()]

        def progress(done, n_local):
            if rank == 0 and time.perf_counter() - last[0] >= 45.0:
                last[0] = time.perf_counter()
                log("sweep: rank 0 has attacked {} of its {} samples".format(done, n_local))

        scratch = None
        if args.from_uint8:
            import tempfile
            scratch = tempfile.mkdtemp(prefix="vqa_bench_attack_dir_")

        def make_source(n, seed):
            from vqattack_amd.attack import dataset
            kind = dataset.SyntheticUint8Pairs if args.from_uint8 else dataset.SyntheticPairs
            return kind(n, text_len, cfg.image_size, flavor, seed=seed, joint=True, dual_every=args.dual_every)

        def one_sweep(n, seed, source=None, as_rank=None):
            r, w = (rank, world) if as_rank is None else (0, 1)
            return run_sweep(flavor, white, black, adapters, n, args.batch, cfg.image_size, text_len, device, r,
                             w, joint=True, seed=seed, log_every=0, save_dir=scratch,
                             dual_every=args.dual_every, mixed=True, attack=sweep_attack,
                             force_collective=use_dist and as_rank is None,
                             collective_device=coll_device, progress=progress,
                             source=source if source is not None else make_source(n, seed))

        if args.emulate_world:
            # ---- the W shards of the fixed set, one after another on this ONE GPU: a prediction, not a measurement
            if world != 1:
                raise SystemExit("--emulate-world runs in one process on one GPU (--gpus 1)")
            from vqattack_amd.attack.asr import shard_indices
            from vqattack_amd.attack.dataset import PairSubset
            W = args.emulate_world
            if args.warmup:
                one_sweep(args.warmup * args.batch, seed=977)
                log("warm-up sweep of {} samples done".format(args.warmup * args.batch))
            full = make_source(args.sweep, 1)
            fence()
            t0 = time.perf_counter()
            shards = []
            for r in range(W):
                shards.append(one_sweep(args.sweep, 1, source=PairSubset(full, shard_indices(args.sweep, r, W)), as_rank=r))
                log("emulated rank {} of {}: {} samples in {:.1f} s".format(r, W, shards[-1]["n_local"], shards[-1]["seconds"]))
            fence()
            dt = time.perf_counter() - t0
            full.close()
            secs = [s["seconds"] for s in shards]
            n_ok = sum(s["asr"] * s["n_total"] for s in shards)
            keys = ("seconds", "n_total", "n_batches", "gradient_steps", "global_steps", "gather_seconds", "n_dual_local",
                    "input_seconds", "input_blocked_seconds", "writer_seconds")
            records = [[s[k] for k in keys] for s in shards]
            line = sweep_line(args, W, [dict(shards[-1], asr=n_ok / args.sweep, collectives=0)], dt, records,
                              [(1, 1)] * W, None, flavor, text_len, cfg.image_size, gemms_tuned=bool(gemms_tuned))
            line.update({
                "value": round(args.sweep / dt, 4), "ms_per_step": round(dt * 1e3, 2), "n_gpus": 1, "steps": 1,
                "scaling": "strong-emulated", "seconds": round(dt, 3), "device_rank0": identity, "distinct_devices": 1,
                "emulated": {
                    "world": W,
                    "what": "the {} rank::{} shards of the fixed set run ONE AFTER ANOTHER on one GPU; ranks of the real "
                            "sweep exchange nothing before the final gathers, so a rank's time there is its shard's time "
                            "here (same GPU model, no contention assumed).  A PREDICTION from the shard imbalance -- NOT a "
                            "multi-GPU measurement; 'value' is this one GPU's throughput over the whole set".format(W, W),
                    "shard_seconds_max": round(max(secs), 3), "shard_seconds_mean": round(sum(secs) / W, 3),
                    "predicted_parallel_seconds": round(max(secs), 3),
                    "predicted_value_at_world": round(args.sweep / max(secs), 4),
                    "predicted_strong_scaling_efficiency": round(sum(secs) / W / max(secs), 4),
                    "gather_seconds_single_process": [round(s["gather_seconds"], 5) for s in shards]}})
            line["config"]["workload"] += " [emulated: {} shards in sequence on 1 GPU]".format(W)
            print(json.dumps(line), flush=True)
            if scratch:
                import shutil
                shutil.rmtree(scratch, ignore_errors=True)
            return
        if args.warmup:
            one_sweep(args.warmup * args.batch * world, seed=977)
            log("warm-up sweep of {} samples done".format(args.warmup * args.batch * world))
        fence()
        t0 = time.perf_counter()
        results = [one_sweep(args.sweep, seed=1) for _ in range(args.steps)]
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=coll_device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        keys = ("seconds", "n_local", "n_batches", "gradient_steps", "global_steps", "gather_seconds", "n_dual_local",
                "input_seconds", "input_blocked_seconds", "writer_seconds")
        record = [results[-1][k] if k in ("n_local", "n_dual_local") else sum(r[k] for r in results) for k in keys]
        records, idents = gather_rank_records(record, identity, use_dist, coll_device)
        if scratch:
            import shutil
            shutil.rmtree(scratch, ignore_errors=True)
        if rank == 0:
            line = sweep_line(args, world, results, dt, records, idents, dist.get_backend() if use_dist else None,
                              flavor, text_len, cfg.image_size, gemms_tuned=bool(gemms_tuned))
            line["seconds"] = round(dt, 3)
            line["device_rank0"] = identity
            print(json.dumps(line), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    ledger = SuccessLedger(world, rank, coll_device, force_collective=use_dist)

    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    images = torch.empty(args.batch, 3, cfg.image_size, cfg.image_size, device=device).uniform_(-1, 1, generator=gen)
    # questions of n ~ U{4..12} words; a joint attack needs its --joint substitutable words
    ids, masks, n_words = synthetic_questions(args.batch, text_len, seed=100 + rank, device=device,
                                              min_words=max(4, args.joint), max_words=max(12, args.joint))
    n_body = max(n_words)
    words = torch.zeros_like(ids, dtype=torch.bool)              # configs[1]: image-only 40-step PGD
    if args.joint:
        words[:, 1:1 + min(args.joint, text_len - 2)] = True     # configs[4]: joint image + text attack
    clean_answers = black.vqa_answer(images, ids, masks)
    tasks = None
    if args.dual:           # every sample's victim answer occurs in its paraphrase -> old_alg == 0 (adv_attack.py:455-469)
        from vqattack_amd.attack.sweep import synthetic_mlm_tasks
        tasks = synthetic_mlm_tasks(ids.cpu(), 1, flavor, seed=rank, max_len=text_len if flavor == "vlmo" else None)
        assert all(t.old_alg == 0 for t in tasks)

    def one_step():
        if tasks is not None:
            res = attack.attack_batch(images, ids, masks, words, dual=True, tasks=tasks)
        else:
            res = attack.attack_batch(images, ids, masks, words)
        adv_answers = black.vqa_answer(res.adv_images, res.adv_text_ids, masks)
        ledger.record(adv_answers != clean_answers)     # sample ids default to this rank's interleaved shard
        return res

    for i in range(args.warmup):
        one_step()
        torch.cuda.synchronize()
        log("warmup step {} done".format(i))
    ledger.reset()
    timer = KernelTimer()
    timer.install()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    asr = ledger.all_gather_rate(world * args.batch * args.steps)   # RCCL all-gather of the success bits (N > 1); inside the timed region
    fence()
    dt = time.perf_counter() - t0
    timer.remove()
    if use_dist:
        t = torch.tensor([dt], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    roof, roof_loss = timer.summary()
    if rank == 0:
        log("timed region: {} steps in {:.2f} s".format(args.steps, dt))

    if rank == 0:
        total = world * args.batch * args.steps
        line = {
            "metric": "adversarial_vqa_examples_per_sec", "value": round(total / dt, 4), "unit": "examples/s",
            "n_gpus": dist.get_world_size() if use_dist else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "{} VQAttack {} ({}): batch {} per GPU, {} PGD steps, {}x{} images, questions of "
                                   "{}..{} real tokens ([CLS] + n words + [SEP], n ~ U{{{}..{}}} seeded) padded to {}{}, "
                                   "eps 0.125 step 0.01 L-inf clip [-1,1], random start, sanity_checks on, black-box "
                                   "scoring + ASR gather".format(
                                       args.model, ("joint image+text attack ({} words)".format(args.joint)
                                                    if args.joint else "image PGD") +
                                       ((", dual loss (feature + MLM step per iteration, MLM head on {})".format(
                                           "all positions" if args.dense_mlm else "the live label rows"))
                                        if args.dual else ""), baseline_config(args), args.batch,
                                       args.pgd_steps, cfg.image_size, cfg.image_size, min(n_words) + 2, n_body + 2,
                                       max(4, args.joint), max(12, args.joint), text_len,
                                       " (the all-padding columns are not run through the encoder)"
                                       if flavor == "vlmo" else ""),
                       "batch_per_gpu": args.batch, "pgd_steps": args.pgd_steps, "image_size": cfg.image_size,
                       "text_len": text_len, "real_tokens": [min(n_words) + 2, n_body + 2],
                       "mean_real_tokens": round(sum(n_words) / len(n_words) + 2, 2), "substitutable_words": args.joint,
                       "sharding": "independent batches per rank, all-gather of success bits"},
            "attack_success_rate": asr,
            "device_rank0": identity,
            "distinct_devices": count_distinct(idents), # GPUs (by UUID / PCI address) the ranks of this line computed on
            "tuned_gemms": ({"file": os.path.relpath(os.environ.get("VQA_TUNED_GEMMS", tuned_gemms.DEFAULT_FILE), ROOT),
                             **tuned_gemms.status()} if gemms_tuned else None),
            "collective": ({"backend": dist.get_backend(), "world": dist.get_world_size(),
                            "tensor_device": str(coll_device), "calls": ledger.collectives}
                           if use_dist else None),
            "roofline": roof,
            "roofline_loss": roof_loss,
        }
        # auxiliary legs: whatever goes wrong in one of them (absent or stale probe library, out of memory on a shared
        # box, ...) is reported inside the line -- never allowed to suppress the metric line or to strand the other ranks
        if not args.no_b256:
            try:
                line["platform_stream_probe"] = stream_probe()
            except Exception as exc:
                line["platform_stream_probe"] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}
            try:
                line["roofline_b256"] = step_kernel_microbench(256, cfg.image_size)
            except Exception as exc:
                line["roofline_b256"] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}
        if cfg.dim // cfg.heads == 64:                # the white box's attention runs on csrc/attn.hip
            seq = (n_body + 2 if flavor == "vlmo" else 0) + cfg.n_image_tokens
            try:
                line["roofline_attention"] = attention_microbench(args.batch, cfg.heads, seq, flavor == "vlmo")
            except Exception as exc:
                line["roofline_attention"] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}
        if not args.no_reference_style and args.model.endswith("_base"):
            try:
                line["reference_style"] = reference_style_leg(flavor, white, cfg, text_len, args.pgd_steps)
            except Exception as exc:
                line["reference_style"] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}
        if world == 1 and not args.no_cpu_baseline and flavor == "vlmo":
            try:
                line["cpu_baseline"] = cpu_baseline(args, cfg)
            except Exception as exc:
                line["cpu_baseline"] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
