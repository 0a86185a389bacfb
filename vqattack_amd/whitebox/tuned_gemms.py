"""Pre-selected library solutions (hipBLASLt or rocBLAS) for the white boxes' fp32 GEMMs (PyTorch TunableOp, read-only).

The attack's device time is ~72 % library fp32 GEMMs (the frozen encoders' projections and expert FFNs, forward and
input-gradient backward: ``torch.addmm`` / ``torch.mm`` in ``whitebox/_fused.py``).  The library's default heuristic picks
one solution per shape; for a few of the attack's shapes another solution (of hipBLASLt, or rocBLAS's own kernel: the
recorded file holds both kinds, `Gemm_Hipblaslt_*` and `Gemm_Rocblas_*`) is faster.
``tools/tune_gemms.sh`` lets TunableOp time every solution for the shapes of a workload on an MI355X and records the
winners; the result (a CSV of shape -> solution index, with the PyTorch / ROCm / hipBLASLt / gfx versions it is valid for)
is tracked under ``vqattack_amd/tuning/``.  ``enable()`` loads it with tuning switched OFF: shapes in the file run the
recorded solution, every other shape the library default; a file recorded for another software stack fails TunableOp's
validators and is ignored (library defaults everywhere) -- so this is never a correctness dependency, and the
arithmetic stays the library's fp32 GEMM.  Every recorded entry is checked on the GPU for run-to-run bit stability and
against the default solution (``tests/test_tuned_gemms.py``), and a white-box attack step with the file active is
required to be bitwise reproducible.
"""
import os

import torch

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning",
                            "tunableop_mi355x_rocm72.csv")


def enable(path=None):
    """Use the recorded GEMM solutions in this process.  Returns True when the file was found and accepted."""
    path = path or os.environ.get("VQA_TUNED_GEMMS", DEFAULT_FILE)
    if path in ("0", "off") or not os.path.exists(path):
        return False
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(False)             # read-only: never time solutions (or write a file) inside an attack
    if hasattr(tunable, "write_file_on_exit"):
        tunable.write_file_on_exit(False)
    else:                                    # whatever the runtime writes at exit must not touch the tracked file
        import tempfile
        tunable.set_filename(os.path.join(tempfile.gettempdir(), "vqa_tunableop_discard_{}.csv".format(os.getpid())))
    ok = bool(tunable.read_file(path))
    if not ok:
        tunable.enable(False)
    return ok


def status():
    import torch.cuda.tunable as tunable
    return dict(enabled=bool(tunable.is_enabled()), tuning=bool(tunable.tuning_is_enabled()),
                entries=len(tunable.get_results()) if tunable.is_enabled() else 0)
