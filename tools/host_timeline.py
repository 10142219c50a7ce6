#!/usr/bin/env python3
"""Where the HOST spends a training step (the small batches are launch-bound: the device waits wherever the host is late).
Wall-clock between the phases of bench.run_step, host side only (nothing here synchronises with the device except the one
synchronisation that ends the timed region): next batch | forward (graph launch) | zero_grad | backward (graph launches) |
clip_grad_norm_ | optimizer.step + scheduler.  usage: python tools/host_timeline.py [batch, default 1] [steps, default 30]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodalsum_amd import optim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
args = bench.parse(["--batch", str(B), "--steps", str(steps), "--warmup", "3", "--no-cpu-baseline", "--no-kernel-probe", "--no-also"])
device = torch.device("cuda", 0)
cfg, model = bench.build(args, device)
opt = optim.get_optimizer(1e-5, bench.NO_DECAY, model.named_parameters(), None)
sch = optim.get_linear_schedule_with_warmup(opt, 100, 100000)
next_batch = bench.batch_source(args, cfg, device, 0)
model.enable_step_graphs()
for _ in range(5):
    bench.run_step(args, model, opt, sch, next_batch())
torch.cuda.synchronize()
names = ["next batch", "forward", "zero_grad", "backward", "clip_grad_norm_", "optimizer + scheduler"]
acc = [0.0] * len(names)
t_begin = time.perf_counter()
for _ in range(steps):
    t = [time.perf_counter()]
    b = next_batch(); t.append(time.perf_counter())
    loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"], b["field_value"], b["img"], b["img_mask"])[0]; t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    optim.clip_grad_norm_(model.parameters(), 1.0, fused=True); t.append(time.perf_counter())
    opt.step(); sch.step(); t.append(time.perf_counter())
    for i in range(len(names)):
        acc[i] += t[i + 1] - t[i]
t_issued = time.perf_counter()
torch.cuda.synchronize()
t_end = time.perf_counter()
print("B = %d: %.2f ms per step on the wall (%d steps); the host had issued everything after %.2f ms per step" %
      (B, (t_end - t_begin) / steps * 1e3, steps, (t_issued - t_begin) / steps * 1e3))
for n, a in zip(names, acc):
    print("  %-24s %7.3f ms per step on the host" % (n, a / steps * 1e3))
