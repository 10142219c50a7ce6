"""Optimiser side of the training step over the flat arena.

  FusedAdamW                     <- HF AdamW, /root/reference/src/transformer/optimization.py:208-267
  clip_grad_norm_                <- torch.nn.utils.clip_grad_norm_ as called at multimodal_train.py:361-362
  get_optimizer                  <- /root/reference/src/train_utils.py:49-57 (quirk Q1 reproduced by default)
  get_linear_schedule_with_warmup<- /root/reference/src/transformer/optimization.py:70-96
"""
import math

import torch
from torch.optim.lr_scheduler import LambdaLR

from . import kernels as kn
from .arena import NO_DECAY


def _arena_of(params):
    for p in params:
        a = getattr(p, "_mmsum_arena", None)
        if a is not None:
            return a
    return None


def tag_arena(engine):
    """Let optimiser utilities find the arena (and offsets) from any of its parameters."""
    a = engine.arena
    for name, p in a.params.items():
        p._mmsum_arena = a
        p._mmsum_name = name
        p._mmsum_engine = engine


def _ranges(arena, params):
    """Merge the arena slices of `params` (those with a gradient) into maximal contiguous ranges."""
    spans = []
    for p in params:
        if p.grad is None:
            continue
        o = arena.offsets[p._mmsum_name]
        n = (arena.numel(p._mmsum_name) + 63) // 64 * 64
        spans.append((o, o + n))
    spans.sort()
    out = []
    for s, e in spans:
        if out and s == out[-1][1]:
            out[-1][1] = e
        else:
            out.append([s, e])
    return out


def clip_grad_norm_(parameters, max_norm, fused=False):
    """Global L2 norm over every parameter that carries a gradient, then scale by
    min(1, max_norm/(norm+1e-6)).  One L2 kernel per contiguous arena range instead of one torch
    op per tensor.  fused=True leaves the scaling of the optimiser-owned gradients to
    FusedAdamW.step() (applied on the fly) and only scales, in place, the ranges the optimiser
    does not own -- under quirk Q1 those are the no-decay gradients that keep accumulating (Q1b),
    so their stored values must carry the clip factor exactly like the reference's do.
    Returns the total norm as a 0-d device tensor (no host sync)."""
    parameters = list(parameters)
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return torch.zeros((), device=parameters[0].device if parameters else None)
    arena = _arena_of(params)
    # Parameters outside the arena (a head or an adapter a user added; a second model's parameters -- the reference clips arbitrary
    # parameter lists with torch's clip_grad_norm_, multimodal_train.py:362, img_pretrain.py:192): their f32 gradients go through the
    # same two library kernels as flat buffers -- the squared norm is added to the arena's, the clip factor scales them in place.
    inside = [p for p in params if arena is not None and getattr(p, "_mmsum_arena", None) is arena]
    foreign = [p for p in params if not (arena is not None and getattr(p, "_mmsum_arena", None) is arena)]
    for p in foreign:
        if not (p.grad.dtype == torch.float32 and p.grad.is_contiguous()):
            raise ValueError("multimodalsum_amd.clip_grad_norm_: a parameter outside the arena needs a contiguous f32 gradient "
                             "(this path has no torch-op fallback)")
    if arena is not None:
        eng = inside[0]._mmsum_engine
        if not hasattr(eng, "norm_sq"):
            eng.norm_sq = torch.zeros(1, device=arena.device)
        norm_sq = eng.norm_sq
        rs = _ranges(arena, inside)
    else:
        eng, rs = None, []
        norm_sq = torch.zeros(1, device=foreign[0].grad.device)
    for i, (s, e) in enumerate(rs):
        kn.l2norm_sq(arena.grad[s:e], norm_sq, accumulate=i > 0)
    for j, p in enumerate(foreign):
        kn.l2norm_sq(p.grad.view(-1), norm_sq, accumulate=(j > 0 or bool(rs)))
    for p in foreign:
        kn.scale_by_clip(p.grad.view(-1), norm_sq, float(max_norm))
    if eng is None:
        return norm_sq.sqrt().reshape(())
    eng.pending_clip = float(max_norm)
    owned = getattr(eng, "optimizer_ranges", None) if fused else None
    for s, e in rs:
        if owned is None:
            kn.scale_by_clip(arena.grad[s:e], eng.norm_sq, float(max_norm))
        else:
            for s2, e2 in _subtract((s, e), owned):
                kn.scale_by_clip(arena.grad[s2:e2], eng.norm_sq, float(max_norm))
    if owned is None:
        eng.pending_clip = None
    return eng.norm_sq.sqrt().reshape(())


def _subtract(span, owned):
    s, e = span
    out = []
    for os_, oe in sorted(owned):
        if oe <= s or os_ >= e:
            continue
        if os_ > s:
            out.append((s, os_))
        s = max(s, oe)
    if s < e:
        out.append((s, e))
    return out


class FusedAdamW(torch.optim.Optimizer):
    """HF AdamW semantics (eps outside the sqrt, bias correction folded into the step size, decoupled
    decay applied after the Adam update with the updated weight) as one kernel per contiguous
    arena range; also writes the bf16 weight shadow and applies a pending gradient clip on the fly.

    The moments live in two flat f32 buffers laid out like the parameter arena; `self.state[p]` holds
    per-parameter VIEWS of them in the HF layout ({"step", "exp_avg", "exp_avg_sq"}, optimization.py:225-232),
    so `optimizer.state_dict()` -- what train_utils.py:97 writes to training_state.bin -- carries the full
    optimiser state and `load_state_dict()` restores it (moments copied back into the flat buffers).
    All parameters of a group that carry a gradient step together (one counter per group, exported per
    parameter): that is what happens in the reference's loop, where every optimised parameter receives a
    gradient in every step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super().__init__(params, defaults)
        self._state_bufs = None
        self._hyper = {}
        self._steps = {}
        self._state_members = {}        # group index -> ids of the parameters whose state views exist

    def _arena(self):
        for g in self.param_groups:
            a = _arena_of(g["params"])
            if a is not None:
                return a
        return None

    def _buffers(self, arena):
        if self._state_bufs is None:
            self._state_bufs = (torch.zeros_like(arena.data), torch.zeros_like(arena.data))
        return self._state_bufs

    def _attach_state(self, arena, gi, group):
        """state[p] = views of the flat moment buffers for every parameter of the group that has a gradient."""
        m, v = self._buffers(arena)
        seen = self._state_members.setdefault(gi, set())
        for p in group["params"]:
            if p.grad is None or id(p) in seen:
                continue
            name = p._mmsum_name
            self.state[p] = {"step": 0, "exp_avg": arena.view(m, name), "exp_avg_sq": arena.view(v, name)}
            seen.add(id(p))

    def _sync_steps(self):
        for gi, group in enumerate(self.param_groups):
            step = self._steps.get((gi,), 0)
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] = step

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)           # torch casts / copies the saved tensors next to their parameters
        arena = self._arena()
        if arena is None:
            return
        m, v = self._buffers(arena)
        self._state_members = {}
        for gi, group in enumerate(self.param_groups):
            steps = []
            for p in group["params"]:
                st = self.state.get(p)
                if not st:
                    continue
                name = p._mmsum_name
                mv, vv = arena.view(m, name), arena.view(v, name)
                mv.copy_(st["exp_avg"].reshape(mv.shape))
                vv.copy_(st["exp_avg_sq"].reshape(vv.shape))
                st["exp_avg"], st["exp_avg_sq"] = mv, vv
                steps.append(int(st["step"]))
                self._state_members.setdefault(gi, set()).add(id(p))
            if steps:
                if min(steps) != max(steps):
                    raise ValueError("FusedAdamW: parameters of group %d were saved at different step counts (%d..%d); the fused "
                                     "update steps a group together" % (gi, min(steps), max(steps)))
                self._steps[(gi,)] = steps[0]

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        arena = self._arena()
        if arena is None:
            raise RuntimeError("FusedAdamW only drives parameters that live in a multimodalsum_amd arena")
        eng = next(p for g in self.param_groups for p in g["params"])._mmsum_engine
        m, v = self._buffers(arena)
        pending = getattr(eng, "pending_clip", None)
        all_ranges = []
        for gi, group in enumerate(self.param_groups):
            rs = _ranges(arena, group["params"])
            if not rs:
                continue
            key = (gi,)
            n_grad = sum(1 for p in group["params"] if p.grad is not None)
            if len(self._state_members.get(gi, ())) != n_grad:
                self._attach_state(arena, gi, group)
            step = self._steps.get(key, 0) + 1
            self._steps[key] = step
            b1, b2 = group["betas"]
            step_size = group["lr"]
            if group["correct_bias"]:
                step_size = step_size * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
            hyper = self._hyper.get(gi)
            if hyper is None:
                hyper = torch.zeros(4, device=arena.device)
                self._hyper[gi] = hyper
            hyper.copy_(torch.tensor([step_size, group["lr"] * group["weight_decay"], pending if pending else 0.0, 0.0]))
            for s, e in rs:
                kn.adamw(arena.data[s:e], arena.grad[s:e], m[s:e], v[s:e], arena.shadow[s:e] if arena.shadow is not None else None,
                         hyper, eng.norm_sq if pending else None, b1, b2, group["eps"])
            all_ranges += rs
        eng.optimizer_ranges = [tuple(r) for r in all_ranges]
        eng.pending_clip = None
        if arena.shadow is not None:
            # parameters outside the updated ranges did not change, so their shadow is still current
            eng.after_fused_optimizer_step()
        return None


def get_optimizer(lr, no_decay, named_parameters, special_condition=None, reproduce_q1=True, fused=True):
    """train_utils.get_optimizer.  reproduce_q1=True keeps the reference's behaviour: the
    `named_parameters` generator is exhausted by the first comprehension, so the no-decay group is
    empty and biases / LayerNorm / BatchNorm weights are never updated."""
    if special_condition is None:
        special_condition = lambda n: True  # noqa: E731
    it = named_parameters if reproduce_q1 else list(named_parameters)
    groups = [
        {'params': [p for n, p in it if special_condition(n) and (not any(nd in n for nd in no_decay))], 'weight_decay': 0.01},
        {'params': [p for n, p in it if special_condition(n) and (any(nd in n for nd in no_decay))], 'weight_decay': 0.0},
    ]
    if fused:
        return FusedAdamW(groups, lr=lr)
    raise ValueError("only the fused optimiser ships with this package; pass the groups to any torch optimiser yourself")


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    def lr_lambda(current_step):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))

    return LambdaLR(optimizer, lr_lambda, last_epoch)
