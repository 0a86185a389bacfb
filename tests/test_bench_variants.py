"""bench.py's workload switches on the test-sized encoders: every variant must produce ONE well-formed JSON line
(metric / unit / roofline objects) -- the dual-loss attack (``--dual``, live-rows and ``--dense-mlm`` forms), the joint
attack (``--joint``), both flavors.  Runs ``bench.main()`` in this process (no launcher, N = 1)."""
import json
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("argv", [
    ["--model", "vlmo_tiny", "--dual", "--joint", "2"],
    ["--model", "vlmo_tiny", "--dual", "--dense-mlm"],
    ["--model", "albef_tiny", "--dual"],
    ["--model", "albef_tiny", "--joint", "2"],
], ids=["vlmo-dual-joint", "vlmo-dual-dense", "albef-dual", "albef-joint"])
def test_bench_variant_prints_one_wellformed_line(argv, capsys, monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "1", "--warmup", "1", "--batch", "4", "--pgd-steps", "8",
                                      "--no-cpu-baseline", "--no-b256"] + argv)
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["metric"] == "adversarial_vqa_examples_per_sec" and rec["unit"] == "examples/s" and rec["n_gpus"] == 1
    assert rec["value"] > 0 and rec["higher_is_better"] is True and rec["dtype"] == "f32" and rec["vs_baseline"] is None
    assert 0.0 <= rec["attack_success_rate"] <= 1.0
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["achieved"] > 0 and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["traffic"] is None               # no PMC phase was recorded for this toy shape
    assert rec["collective"] is None
    dual = "--dual" in argv
    assert ("dual loss" in rec["config"]["workload"]) == dual
    n_words = int(argv[argv.index("--joint") + 1]) if "--joint" in argv else 0
    assert roof["launches"] >= 4                  # fused steps were timed inside the timed region
    assert rec["config"]["substitutable_words"] == n_words
