"""Where do the bf16 ResNet paths (timed: epilogue statistics + implicit convolutions; deterministic: separate reductions + im2col) part,
and which one is closer to the f32 engine?  Per BatchNorm layer in forward order: relative difference of the batch mean / variance."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodalsum_amd.modules import MultimodalSum
from tests.test_host_logic_cpu import tiny_cfg
cfg = tiny_cfg(vocab=60, d=256, ffn=64, layers=1, heads=4, maxpos=40)
g = torch.Generator().manual_seed(5)
img = torch.randn(6, 3, 96, 96, generator=g).cuda()
runs = {}
for name, dt, det, env in (("f32", torch.float32, True, None), ("bf16_det", torch.bfloat16, True, None), ("bf16_timed", torch.bfloat16, False, None),
                           ("bf16_timed_im2col", torch.bfloat16, False, "0")):
    if env is not None:
        os.environ["MMSUM_IMPLICIT_CONV"] = env
    else:
        os.environ.pop("MMSUM_IMPLICIT_CONV", None)
    model = MultimodalSum(config=cfg, label_smoothing=0.1, device="cuda", dtype=dt, deterministic=det)
    e = model._engine
    e.sync_weights(); e.arena.prepare_grads()
    y, c = e.img_fwd(img)
    torch.cuda.synchronize()
    runs[name] = (y.double().cpu(), {k: v.double().cpu().clone() for k, v in e.buffers.items() if "running" in k})
keys = [k for k in runs["f32"][1] if k.endswith("running_mean")]
def stat(run, k):
    m = runs[run][1][k] * 10
    v = (runs[run][1][k.replace("running_mean", "running_var")] - 0.9) * 10
    return m, v
print("%-52s %s" % ("layer", "  ".join("%-26s" % n for n in ("bf16_det vs f32", "bf16_timed vs f32", "timed_im2col vs f32", "timed vs det"))))
for k in keys:
    cells = []
    for a_, b_ in (("bf16_det", "f32"), ("bf16_timed", "f32"), ("bf16_timed_im2col", "f32"), ("bf16_timed", "bf16_det")):
        ma, va = stat(a_, k); mb, vb = stat(b_, k)
        cells.append("m %.1e v %.1e" % (float((ma - mb).abs().max() / (mb.abs().max() + 1e-9)), float((va - vb).abs().max() / (vb.abs().max() + 1e-9))))
    print("%-52s %s" % (k.replace("img_encoder.resnet.", "").replace(".running_mean", ""), "  ".join("%-26s" % c for c in cells)))
yf = runs["f32"][0]
for n in ("bf16_det", "bf16_timed", "bf16_timed_im2col"):
    print(n, "output vs f32: rel L2 %.3e" % float((runs[n][0] - yf).norm() / yf.norm()))
