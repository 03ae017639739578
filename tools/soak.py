"""Soak / sanity run: N full training steps on ONE fixed 8+8-scene synthetic batch.  The loss must stay finite and fall
(the nets can memorise a single batch).  Every reduction of the step is order-fixed, so two runs must print IDENTICAL
losses to the last digit: `diff` of two outputs is the race detector for the pipelined kernels (counted waits, LDS rings)."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
from mm2d3d_amd.synthetic import make_batch  # noqa: E402
tm = bench.build_trainer(dev, total_steps=steps + 10)
batch = {"source": make_batch(2, 8, "nuscenes", (302, 480), 6, device=dev, augment=True),
         "target": make_batch(3, 8, "nuscenes", (302, 480), 6, device=dev, augment=True)}
losses = []
for i in range(steps):
    loss = tm.fit_step(bench.fresh(batch))
    if i % 10 == 0 or i == steps - 1:
        v = float(loss)
        losses.append(v)
        print(f"step {i:4d} loss {v:.9g}", flush=True)
        assert math.isfinite(v), "non-finite loss"
assert losses[-1] < 0.6 * losses[0], f"loss did not fall: {losses[0]:.3f} -> {losses[-1]:.3f}"
print("OK", losses[0], "->", losses[-1])
