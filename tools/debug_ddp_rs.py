import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
import ddp_rccl_worker as w
from multimodalsum_amd.parallel import DistributedDataParallel
res = {}
chunks = {}
_orig = DistributedDataParallel._all_reduce_mean
def _logged(self, chunk):
    g = self.arena.grad
    off = (chunk.data_ptr() - g.data_ptr()) // 4
    chunks.setdefault(self._tag, []).append((off, chunk.numel()))
    return _orig(self, chunk)
DistributedDataParallel._all_reduce_mean = _logged
for mode, bucket in (("none", 0), ("all_reduce", 1 << 20), ("reduce_scatter", 1 << 20), ("reduce_scatter_nooverlap", 1 << 20)):
    cfg, model = w.build(torch.float32, dev)
    kw = dict(overlap=False) if mode.endswith("nooverlap") else {}
    runner = model if mode == "none" else DistributedDataParallel(model, delay_allreduce=True, always_reduce=True, collect_stats=not mode.endswith("nostats"), bucket_elems=bucket,
                                                                   mode=mode.split("_no")[0], **kw)
    if mode != "none":
        runner._tag = mode
    arena = model._engine.arena
    for p in model.parameters():
        p.grad = None
    loss = w.step(runner, w.batch(cfg, 0, dev))
    torch.cuda.synchronize()
    res[(mode, bucket)] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
base = res[("none", 0)]
for key, g in res.items():
    if key[0] == "none":
        continue
    bad = sorted(((float((g[n] - base[n]).abs().max()), n) for n in base), reverse=True)[:6]
    print(key, "worst:", [(round(e, 6), n) for e, n in bad if e > 0] or "identical")
    for e, n in bad[:2]:
        if e > 0:
            d = (g[n] - base[n]).reshape(-1)
            nz = d.nonzero().reshape(-1)
            big = (d.abs() > 1e-6).nonzero().reshape(-1)
            poff = (dict(model.named_parameters())[n].grad.data_ptr() - arena.grad.data_ptr()) // 4
            print("    LARGE errors at", big.tolist()[:20], "of", big.numel(), "; param arena offset", poff, "numel", d.numel(), "diffs", d[big[:8]].tolist())
            near = [(o, l) for o, l in chunks.get(key[0], []) if abs(o - (poff + d.numel())) < 4096 or abs(o + l - poff) < 4096 or (o <= poff < o + l)]
            print("    chunks touching / next to it (offset, len, in call order index):", [(o, l, chunks[key[0]].index((o, l))) for o, l in near])
            print("   ", n, "differs at", nz.numel(), "of", d.numel(), "elements; first", nz[:4].tolist(), "rs", g[n].reshape(-1)[nz[:4]].tolist(), "base", base[n].reshape(-1)[nz[:4]].tolist())
dist.destroy_process_group()
