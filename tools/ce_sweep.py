import sys, torch
sys.path.insert(0, '.')
from vqattack_amd import ops, _hip
rows, v = 2560, 30522
logits = torch.randn(rows, v, device="cuda"); labels = torch.randint(0, v, (1, rows), device="cuda"); slot = torch.zeros(1, device="cuda")
for thr in (256, 512, 1024, 256, 512, 1024):
    assert _hip.lib().vqa_set_option(4, thr) == 0
    for _ in range(3): ops.mlm_cross_entropy(logits, labels, slot, accumulate=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.mlm_cross_entropy(logits, labels, slot, accumulate=False)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(thr, round(us, 1), "us", round(8 * rows * v / us / 1e3), "GB/s")
