"""Hand-scheduled forward/backward programs of the hot path over the HIP kernels.

There is no tracing compiler and no autograd graph inside the model: each sub-program below
(encoder, table encoder, ResNet, multi-encoder decoder, LM head + loss) is an explicit kernel
schedule with its own saved activations, and its backward is the explicit reverse schedule that
writes parameter gradients straight into the flat gradient arena (arena.py).  torch.autograd only
sees a few coarse nodes (modules.py), so whole steps can be captured into HIP graphs.

Restructuring relative to the reference:
  * the NR leave-one-out decoder passes (multimodal_train.py:150-163) run as ONE decoder call with
    NR*B sequences; the cross-attention kernel skips entity i for pass i (exclude_self);
  * cross-attention K/V of all reviews + table + images are projected once per layer per step
    (one [Rmem, 2D] GEMM), q once per layer, the three out_proj calls as one [3*Rq, D] GEMM;
  * torch.cat([text, table]) for alpha/beta is a K-split GEMM over two A operands.
Numerics follow the reference op for op (same masks, same -inf/-2^16 semantics, entity mean with
null entities, post-LN, erf-GELU); see the citations at each step.
"""
import math
import os
from types import SimpleNamespace as NS

import torch

from . import kernels as kn
from .arena import ParamArena

RESNET_LAYERS = (3, 4, 23, 3)


# ------------------------------------------------------------------------------------------------
# parameter inventory (names = the reference's state_dict keys)
# ------------------------------------------------------------------------------------------------
def bart_specs(cfg, multimodal, prefix):
    D, V = cfg.d_model, cfg.vocab_size
    P = cfg.max_position_embeddings + cfg.extra_pos_embeddings
    s = [(prefix + "model.shared.weight", (V, D))]
    for side, nl, Fd in (("encoder", cfg.encoder_layers, cfg.encoder_ffn_dim), ("decoder", cfg.decoder_layers, cfg.decoder_ffn_dim)):
        b = prefix + "model.%s." % side
        s.append((b + "embed_positions.weight", (P, D)))
        s += [(b + "layernorm_embedding.weight", (D,)), (b + "layernorm_embedding.bias", (D,))]
        if side == "decoder":
            s.append((b + "rating_embeddings", (D,)))
        for i in range(nl):
            lb = b + "layers.%d." % i
            for a in (["self_attn"] + (["encoder_attn"] if side == "decoder" else [])):
                # q,k,v adjacent (weights in the decay group, biases in the no-decay group)
                for p in ("q_proj", "k_proj", "v_proj"):
                    s.append((lb + a + "." + p + ".weight", (D, D)))
                for p in ("q_proj", "k_proj", "v_proj"):
                    s.append((lb + a + "." + p + ".bias", (D,)))
                s += [(lb + a + ".out_proj.weight", (D, D)), (lb + a + ".out_proj.bias", (D,))]
                if a == "encoder_attn" and multimodal:
                    for p in ("alpha_proj", "beta_proj"):
                        s += [(lb + a + "." + p + ".weight", (D, 2 * D)), (lb + a + "." + p + ".bias", (D,))]
                s += [(lb + a + "_layer_norm.weight", (D,)), (lb + a + "_layer_norm.bias", (D,))]
            s += [(lb + "fc1.weight", (Fd, D)), (lb + "fc1.bias", (Fd,)), (lb + "fc2.weight", (D, Fd)), (lb + "fc2.bias", (D,))]
            s += [(lb + "final_layer_norm.weight", (D,)), (lb + "final_layer_norm.bias", (D,))]
    return s


def table_specs(prefix="table_encoder.", kind="yelp"):
    """YelpTableEncoder (table_encoder.py:5-12) or AmazonTableEncoder (:86-93) parameters."""
    first = [(prefix + "rating_embedding.weight", (1024, 4)), (prefix + "hours_embedding.weight", (1024, 4))] if kind == "yelp" else \
            [(prefix + "price_embedding.weight", (1024, 11)), (prefix + "rating_embedding.weight", (1024, 4))]
    return first + [(prefix + "fc.weight", (1024, 2048)), (prefix + "fc.bias", (1024,)), (prefix + "linear.weight", (1024, 1024))]


TABLE_POSITIONS = {"yelp": 47, "amazon": 133}


def resnet_blocks():
    """(layer index, block index, inplanes, planes, stride, has_downsample) of torchvision resnet101."""
    out, inplanes = [], 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), RESNET_LAYERS)):
        for bi in range(blocks):
            out.append((li + 1, bi, inplanes, planes, 2 if (bi == 0 and li > 0) else 1, bi == 0))
            inplanes = planes * 4
    return out


def resnet_specs(embedding_dim, prefix="img_encoder."):
    """Returns (param specs, buffer specs).  Live (layer3 + linear) parameters first, then the
    gradient-free ones (stem/layer1/layer2: detached at img_encoder.py:33; layer4/fc: never called)."""
    r = prefix + "resnet."
    live, frozen, bufs = [], [], []

    def bn(dst, name, c):
        dst += [(name + ".weight", (c,)), (name + ".bias", (c,))]
        bufs.extend([(name + ".running_mean", (c,)), (name + ".running_var", (c,)), (name + ".num_batches_tracked", ())])

    frozen.append((r + "conv1.weight", (64, 3, 7, 7)))
    bn(frozen, r + "bn1", 64)
    for li, bi, inp, pl, stride, down in resnet_blocks():
        dst = live if li == 3 else frozen
        b = r + "layer%d.%d." % (li, bi)
        dst.append((b + "conv1.weight", (pl, inp, 1, 1)))
        bn(dst, b + "bn1", pl)
        dst.append((b + "conv2.weight", (pl, pl, 3, 3)))
        bn(dst, b + "bn2", pl)
        dst.append((b + "conv3.weight", (pl * 4, pl, 1, 1)))
        bn(dst, b + "bn3", pl * 4)
        if down:
            dst.append((b + "downsample.0.weight", (pl * 4, inp, 1, 1)))
            bn(dst, b + "downsample.1", pl * 4)
    frozen += [(r + "fc.weight", (1000, 2048)), (r + "fc.bias", (1000,))]
    live.append((prefix + "linear.weight", (embedding_dim, 1024)))
    return live, frozen, bufs


def splitk_rule(M, N, Kred, bf16=True, deterministic=False, tile256=False):
    """Split count of the weight-gradient product dW[M, N] = dy[Kred, M]^T x[Kred, N] (Engine.wgrad; the bench-shape parity tests
    call this same function, so they run the step's real slice counts).  tile256: the product runs on 256x256 tiles whatever its
    size (mmsum_conv3x3_wgrad)."""
    if deterministic:
        return 1
    if bf16 and not tile256 and Kred < 4096:
        # few rows (the per-GPU batch 1 of multimodal_train.py:420: 1,152 decoder rows): the 128x128 tile list of a Linear's gradient
        # already covers half the CUs or all of them, a slice would hold a handful of 32-deep slabs, and the slab pass is a launch
        # of its own -- tools/gemm_small_bench.py at 1,152 rows: dW[4096,1024] 25.8 us unsplit against 40.3 at 4 slices,
        # dW[3072,1024] 23.6 / 35.9, dW[1024,1024] 19.0 at 2 / 21.4 at 4; at 2,304 rows dW[1024,1024] 24.7 at 4 / 29.5 at 9
        t128 = ((M + 127) // 128) * ((N + 127) // 128)
        return max(1, min(256 // t128, (Kred // 32) // 16))
    ktiles = max(1, Kred // (64 if bf16 else 32))
    # one 256x256 tile per CU and launch: the TN kernel is not persistent, so tiles x slices should come as close to the 256
    # CUs as it can from below (tools/tn_sk_sweep.py at R = 64,512: dW[4096,1024] 521 us at 3 slices = 192 workgroups, 451 us
    # at 4 = 256, 613 us at 5 = 320); few-tile outputs with a very long reduction (the ResNet layer3 convolutions: 4..9 tiles,
    # 43,904 rows) keep gaining up to 32 slices (tools/wgrad_small.py: 77 -> 46 us)
    if not bf16:                                           # f32 parity mode (generic kernel, 128x128 tiles): the round-1 rule
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        return max(1, min(32, -(-768 // max(tiles, 1)), ktiles // 4))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    return max(1, min(32, 256 // tiles if tiles <= 256 else 1, ktiles // 4))


class Engine:
    def __init__(self, cfg, device="cuda", compute_dtype=torch.bfloat16, multimodal=True, with_table=False, with_img=False,
                 bart_prefix="", deterministic=False):
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = compute_dtype
        self.multimodal = multimodal
        self.with_table, self.with_img = bool(with_table), with_img
        self.table_kind = None if not with_table else ("amazon" if with_table == "amazon" else "yelp")
        self.table_positions = TABLE_POSITIONS.get(self.table_kind, 0)
        self.bp = bart_prefix
        self.deterministic = deterministic
        # bf16: stride-1 3x3 convolutions as implicit GEMMs (False / MMSUM_IMPLICIT_CONV=0: im2col + GEMM everywhere; tests and A/B runs compare the two)
        self.implicit_conv = os.environ.get("MMSUM_IMPLICIT_CONV") != "0"
        self.training = True
        self.seed_base = 0x5EED
        self.step_count = 0
        self.seed_log = None
        specs = bart_specs(cfg, multimodal, bart_prefix)
        self.buffers = {bart_prefix + "final_logits_bias": torch.zeros(1, cfg.vocab_size, device=self.device)}
        frozen = []
        if with_table:
            specs += table_specs(kind=self.table_kind)
        if with_img:
            live, frozen, bufs = resnet_specs(cfg.d_model)
            specs += live
            # BatchNorm step counters live in one vector (layers the hot path runs first) so that a forward bumps
            # them with one launch instead of one per layer
            nbt = [n for n, _ in bufs if n.endswith("num_batches_tracked")]
            nbt.sort(key=lambda n: ".layer4." in n)
            self._nbt_all = torch.zeros(len(nbt), dtype=torch.int64, device=self.device)
            self._nbt_live = sum(1 for n in nbt if ".layer4." not in n)
            for i, n in enumerate(nbt):
                self.buffers[n] = self._nbt_all[i]
            for name, shape in bufs:
                if name.endswith("num_batches_tracked"):
                    continue
                else:
                    self.buffers[name] = (torch.ones if name.endswith("running_var") else torch.zeros)(shape, device=self.device)
        self.frozen_names = [n for n, _ in frozen]
        self.arena = ParamArena(specs + frozen, self.device, compute_dtype)
        self.conv_mats = {}
        self._conv_dirty = "all"
        self.touched = set()
        self.Vpad = (cfg.vocab_size + 127) // 128 * 128
        self.wt, self.wt_desc, self.conv_mats_t, self.conv_mats_r, self._conv_graphs = {}, None, {}, {}, {}
        if compute_dtype == torch.bfloat16:
            self._build_wt_table()
        self._ip = None                    # live-image window of the image branch while img_fwd / img_bwd run (kn.ImagePlan or None)
        self.salt = None                   # device uint64 mixed into every dropout seed (set by graphs.StepGraphs; None = seeds as passed)
        self.post_backward_hooks = []      # run once when a whole backward pass has finished (DDP finalisation)
        self.segment_hooks = []            # run when a parameter segment's gradients are final (DDP overlap)
        for name, p in self.arena.params.items():   # lets optim.py / parallel.py find the arena from a parameter
            p._mmsum_arena, p._mmsum_name, p._mmsum_engine = self.arena, name, self

    def segment_ready(self, prefixes):
        """Gradients of every parameter whose name starts with one of `prefixes` are final."""
        for cb in self.segment_hooks:
            cb(prefixes)

    # ---- helpers --------------------------------------------------------------------------------
    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.device)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.dtype, device=self.device)

    def side_stream(self):
        """Second HIP stream for the image/table branch of the fused step (None when MMSUM_SIDE_STREAM=0)."""
        if not hasattr(self, "_side_stream"):
            import os
            self._side_stream = None if os.environ.get("MMSUM_SIDE_STREAM") == "0" else torch.cuda.Stream(device=self.device)
        return self._side_stream

    def p_drop(self):
        return float(self.cfg.dropout) if self.training else 0.0

    def next_seed(self):
        """Seed of the next dropout site (embedding, self-attention, cross-attention, FFN blocks in schedule order).  `seed_log`
        (a list, None = off) records them: with dropout.keep_mask a checker rebuilds every mask of a step on the host."""
        self.step_count += 1
        seed = (self.seed_base * 1000003 + self.step_count) & 0xFFFFFFFFFFFF
        if self.seed_log is not None:
            self.seed_log.append(seed)
        return seed

    # ---- transposed bf16 weight shadows: dgrad dx = dy W runs as the NT product dy (W^T)^T --------
    def _wt_groups(self):
        cfg, bp = self.cfg, self.bp
        D, V = cfg.d_model, cfg.vocab_size
        g = [(bp + "model.shared.weight", bp + "model.shared.weight", V, D, self.Vpad)]
        for side, nl in (("encoder", cfg.encoder_layers), ("decoder", cfg.decoder_layers)):
            for i in range(nl):
                lb = bp + "model.%s.layers.%d." % (side, i)
                q, k, v = self._attn_names(lb, "self_attn")
                g.append((q + ".weight", v + ".weight", 3 * D, D, 3 * D))
                g.append((lb + "self_attn.out_proj.weight", lb + "self_attn.out_proj.weight", D, D, D))
                if side == "decoder":
                    q, k, v = self._attn_names(lb, "encoder_attn")
                    g.append((q + ".weight", q + ".weight", D, D, D))
                    g.append((k + ".weight", v + ".weight", 2 * D, D, 2 * D))
                    g.append((lb + "encoder_attn.out_proj.weight", lb + "encoder_attn.out_proj.weight", D, D, D))
                    if self.multimodal:
                        for pr in ("alpha_proj", "beta_proj"):
                            g.append((lb + "encoder_attn.%s.weight" % pr, lb + "encoder_attn.%s.weight" % pr, D, 2 * D, D))
                Fd = self.arena.shapes[lb + "fc1.weight"][0]
                g.append((lb + "fc1.weight", lb + "fc1.weight", Fd, D, Fd))
                g.append((lb + "fc2.weight", lb + "fc2.weight", D, Fd, D))
        if self.with_table:
            g.append(("table_encoder.fc.weight", "table_encoder.fc.weight", 1024, 2048, 1024))
            g.append(("table_encoder.linear.weight", "table_encoder.linear.weight", 1024, 1024, 1024))
        if self.with_img:
            g.append(("img_encoder.linear.weight", "img_encoder.linear.weight", D, 1024, D))
            for li, bi, inp, pl, stride, down in resnet_blocks():
                if li == 3:
                    b = "img_encoder.resnet.layer3.%d." % bi
                    g.append((b + "conv1.weight", b + "conv1.weight", pl, inp, pl))
                    g.append((b + "conv3.weight", b + "conv3.weight", pl * 4, pl, pl * 4))
        return g

    def _build_wt_table(self):
        a = self.arena
        rows_desc, off, max_tiles = [], 0, 1
        for first, last, rows, cols, ld in self._wt_groups():
            src_off = a.offsets[first]
            assert a.offsets[last] + a.numel(last) - src_off == rows * cols, (first, last)
            rows_desc.append([src_off, off, rows, cols, cols, ld])
            self.wt[first] = (off, cols, ld, rows)
            off += cols * ld
            off = (off + 63) // 64 * 64
            max_tiles = max(max_tiles, ((rows + 63) // 64) * ((cols + 63) // 64))
        self.wt_buf = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self.wt_desc = torch.tensor(rows_desc, dtype=torch.int64, device=self.device)
        self.wt_max_tiles = max_tiles
        for k, (o, cols, ld, rows) in list(self.wt.items()):
            self.wt[k] = self.wt_buf[o:o + cols * ld].view(cols, ld)

    def dgrad(self, dy, key, w_natural, out, accumulate=False, epi=0, aux=None, rows=None, colsum=None, live=None, alpha_dev=None):
        """out (+)= dy @ W  (W natural = [N_out, K_in]); uses the transposed shadow when present.
        colsum (f32 [K_in], optional) += column sums of out: the bias gradient of the layer below, taken in the GEMM
        epilogue when the fast path allows it and by the column-sum kernel otherwise.
        live: device row count of dy / out (compacted rows); alpha_dev: device scalar multiplied into the product."""
        wt = self.wt.get(key)
        if wt is not None:
            fuse = colsum is not None and not accumulate and not self.deterministic and kn.gemm_colsum_fusable(dy)
            kn.gemm(dy, wt if rows is None else wt[rows], out, accumulate=accumulate, epi=epi, aux=aux, colsum=colsum if fuse else None,
                    live=live, alpha_dev=alpha_dev)
            if colsum is not None and not fuse:
                kn.colsum(out, colsum, accumulate=True, live=live)
        else:
            kn.gemm(dy, w_natural, out, b_t=True, accumulate=accumulate, epi=epi, aux=aux, live=live, alpha_dev=alpha_dev)
            if colsum is not None:
                kn.colsum(out, colsum, accumulate=True, live=live)

    def sync_weights(self):
        """Called at the start of every forward.  Parameters are ordinary f32 tensors that any
        optimiser may update in place, so the bf16 shadow and the conv weight matrices are rebuilt
        from the masters unless the fused optimiser (optim.FusedAdamW) has declared them current."""
        a = self.arena
        if a.shadow is not None and a.shadow_dirty:
            kn.cast(a.shadow, a.data)
        if self.with_img and self._conv_dirty:
            self._refresh_conv_mats(self._conv_dirty)
        if self.wt_desc is not None and self.training:
            kn.transpose_batched(a.shadow, self.wt_buf, self.wt_desc, self.wt_desc.shape[0], self.wt_max_tiles)
        a.shadow_dirty = True
        self._conv_dirty = "all"

    def _refresh_conv_mats(self, which):
        """The conv weight matrices from the f32 masters: ~70 launches of a few microseconds each, issued from Python once per step.
        On the GPU the launch sequence is captured once per `which` ("all" / "layer3": pointers and shapes never change) and replayed
        as ONE graph launch (kernel trace at B = 8: 45 gaps of ~15 us per step before these kernels)."""
        all_layers = which == "all"
        if self.device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return self._build_conv_mats(all_layers)
        g = self._conv_graphs.get(which)
        if g is None:
            self._build_conv_mats(all_layers)            # eagerly first: allocates the matrices
            import gc
            torch.cuda.synchronize()
            gc.collect()
            gc_was = gc.isenabled()
            gc.disable()                                 # a CUDAGraph finalised by the collector during capture aborts the capture
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._build_conv_mats(all_layers)
            finally:
                if gc_was:
                    gc.enable()
            self._conv_graphs[which] = g
            return
        g.replay()

    def mark_weights_changed(self):
        self.arena.shadow_dirty = True
        self._conv_dirty = "all"

    def after_fused_optimizer_step(self):
        """FusedAdamW wrote the bf16 shadow itself; only layer3's 3x3 matrices need a refresh."""
        self.arena.shadow_dirty = False
        self._conv_dirty = "layer3"

    def splitk(self, M, N, Kred, tile256=False):
        return splitk_rule(M, N, Kred, bf16=self.dtype == torch.bfloat16, deterministic=self.deterministic, tile256=tile256)

    def wgrad(self, dy, x, gname=None, gview=None, bias_g=None, live=None, alpha_dev=None):
        """dW[N_out, K_in] += dy[R, N_out]^T x[R, K_in] into the f32 gradient arena (live: device count of the rows R).
        bias_g (f32 [N_out], optional) += column sums of dy, the bias gradient of the same Linear: inside the weight-gradient
        kernel where it can carry them (bf16 four-wave TN kernel: the sums come from the operand tiles it stages anyway), by the
        column-sum kernel otherwise (f32 / deterministic mode, small products)."""
        out = gview if gview is not None else self.arena.g(gname)
        R = dy.shape[0]
        sk = self.splitk(dy.shape[1], x.shape[1], R)
        if self.dtype == torch.bfloat16 and sk > 1 and x.shape[1] % 4 == 0:
            # reduction-major product straight from the activations (gemm_tn_ring_kernel: transposing LDS reads);
            # split-K partial slabs + a deterministic reduce: cheaper than f32 atomics (1.3 TB/s chip-wide)
            ws = self.empty(sk * dy.shape[1], x.shape[1], dtype=torch.float32)
            fuse = bias_g is not None and not self.deterministic and kn.gemm_tn_colsum_ok(dy, x, ws, sk, bias_g)
            if bias_g is not None and not fuse:
                kn.colsum(dy, bias_g, accumulate=True, live=live)
            kn.gemm(dy, x, ws, a_t=True, b_t=True, splitk=sk, slabs=True, live=live, alpha_dev=alpha_dev, colsum=bias_g if fuse else None)
            kn.slab_reduce(ws, sk, out, accumulate=True)
            return
        if bias_g is not None:
            kn.colsum(dy, bias_g, accumulate=True, live=live)
        kn.gemm(dy, x, out, a_t=True, b_t=True, accumulate=True, splitk=sk, live=live, alpha_dev=alpha_dev)

    def bgrad(self, dy, gname=None, gview=None, live=None):
        kn.colsum(dy, gview if gview is not None else self.arena.g(gname), accumulate=True, live=live)

    def touch(self, *names):
        self.touched.update(names)

    # =============================================================================================
    # Encoder  (BartEncoder.forward, modeling_multimodalsum.py:346-404; EncoderLayer :276-309)
    # =============================================================================================
    def _attn_names(self, lb, a):
        return [lb + a + "." + p for p in ("q_proj", "k_proj", "v_proj")]

    def row_maps(self, keep):
        """Index maps of the padding-free parts: keep = any tensor whose non-zero entries mark the live rows (row-major).
        c2p [R] = padded row of compact row i (-1 past the live count), p2c [R] = compact row of padded row r (-1 for
        padding; p2c32 = the same as int32: the row map the bf16 attention kernels read compact matrices through),
        count = the live count as a device int32 scalar.  Everything stays on the device and has static shapes:
        the kernels read `count` when they run (their `live_rows` argument), so neither a host read nor a per-count graph
        is needed; the compact buffers are simply sized for all R rows."""
        flat = keep.reshape(-1).ne(0)
        R = flat.numel()
        pos = torch.cumsum(flat.to(torch.int64), 0) - 1
        p2c = torch.where(flat, pos, torch.full_like(pos, -1))
        ar = torch.arange(R, dtype=torch.int64, device=flat.device)
        c2p = torch.full((R + 1,), -1, dtype=torch.int64, device=flat.device)       # slot R swallows the padding rows' writes
        c2p.index_put_((torch.where(flat, pos, torch.full_like(pos, R)),), torch.where(flat, ar, torch.full_like(ar, -1)))
        count = flat.sum().to(torch.int32).reshape(1)
        return NS(c2p=c2p[:R].contiguous(), p2c=p2c.contiguous(), p2c32=p2c.to(torch.int32).contiguous(), count=count, rows=R)

    def encoder_fwd(self, ids, attention_mask, out=None, compact=False):
        """ids [Bn,S] int64, attention_mask [Bn,S] (1 = keep).  -> hidden [Bn*S, D] (batch-major rows).
        compact (fused step only): run the layers' GEMM / LayerNorm work on the valid rows only (rows that are padding
        come out as zeros; nothing downstream reads them: they are masked keys of the cross-attention)."""
        cfg, a = self.cfg, self.arena
        D, H = cfg.d_model, cfg.heads
        Bn, S_in = ids.shape
        # Sequences of more than 128 tokens (test.py:56-60 tokenises Yelp reviews to 158): the attention kernels take query blocks of
        # at most 128 rows, so a sequence is cut into `nsplit` equal query blocks that share the sequence's keys (_self_block_fwd).
        # An odd length gets one more padding column here (pad token, mask 0: a masked key and a query nobody reads -- the valid
        # positions' results do not change, BART's learned positions are absolute) and loses it again on the way out.
        nsplit = self.seq_splits(S_in)
        S = -(-S_in // nsplit) * nsplit
        unpad = None
        if S != S_in:
            ids = torch.nn.functional.pad(ids, (0, S - S_in), value=cfg.pad_token_id)
            attention_mask = torch.nn.functional.pad(attention_mask, (0, S - S_in), value=0)
            unpad = (torch.arange(Bn, device=ids.device).view(Bn, 1) * S + torch.arange(S_in, device=ids.device).view(1, S_in)).reshape(-1)
        R = Bn * S
        b = self.bp + "model.encoder."
        c = NS(Bn=Bn, S=S, S_in=S_in, unpad=unpad, ids=ids.contiguous(), layers=[], p=self.p_drop())
        c.pad = attention_mask.eq(0).to(torch.uint8).contiguous()
        c.seed0 = self.next_seed()
        x = self.empty(R, D)
        c.mean0, c.rstd0 = self.empty(R, dtype=torch.float32), self.empty(R, dtype=torch.float32)
        kn.embed_ln_fwd(c.ids, a.w(self.bp + "model.shared.weight"), a.w(b + "embed_positions.weight"), None, None,
                        a.f32(b + "layernorm_embedding.weight"), a.f32(b + "layernorm_embedding.bias"), x, c.mean0, c.rstd0,
                        Bn, S, cfg.extra_pos_embeddings, 1e-5, c.p, c.seed0, salt=self.salt)
        c.x0 = x
        c.maps = None
        if compact:
            c.maps = self.row_maps(attention_mask)
            x = kn.rows_gather(x, self.empty(R, D), c.maps.c2p, live=c.maps.count)
        for i in range(cfg.encoder_layers):
            last = i == cfg.encoder_layers - 1
            x, lc = self._self_block_fwd(b + "layers.%d." % i, x, c.pad, Bn, S, causal=False, maps=c.maps)
            x, fc = self._ffn_block_fwd(b + "layers.%d." % i, x, out if (last and c.maps is None and unpad is None) else None, maps=c.maps)
            c.layers.append((lc, fc))
        if c.maps is not None:
            x = kn.rows_gather(x, out if out is not None else self.empty(Bn * S_in, D), c.maps.p2c if unpad is None else c.maps.p2c[unpad])
        elif unpad is not None:
            x = kn.rows_gather(x, out if out is not None else self.empty(Bn * S_in, D), unpad)
        c.out = x
        return x, c

    @staticmethod
    def seq_splits(S):
        """Query blocks per sequence of S tokens in the encoder's self-attention (blocks of <= 128 rows over <= 224 keys)."""
        if S > 224:
            raise ValueError("encoder sequences of %d tokens: the self-attention kernels stage at most 224 keys per sequence "
                             "(the reference's inputs are 128 tokens in training and 158 / 118 in test.py)" % S)
        return 1 if S <= 128 else 2

    def _pad_rows(self, c):
        """int64 map [Bn * S]: row of the caller's [Bn * S_in] layout behind every row of the internally padded one (-1 = the padding column)."""
        m = torch.full((c.Bn * c.S,), -1, dtype=torch.int64, device=c.unpad.device)
        m[c.unpad] = torch.arange(c.Bn * c.S_in, dtype=torch.int64, device=c.unpad.device)
        return m

    def encoder_bwd(self, c, dout, split=None):
        """dout [Bn*S, D] (consumed).  Accumulates every encoder parameter gradient.
        split=(lo, hi, carry): run layers hi-1 .. lo only (the fused step cuts the encoder backward in two gradient segments
        so that the data-parallel all-reduce of the upper layers overlaps the lower ones); `carry` is the input gradient
        handed from the upper part (None for the part that starts at the top) and the function returns it for the next."""
        cfg, a = self.cfg, self.arena
        b = self.bp + "model.encoder."
        lo, hi, carry = (0, cfg.encoder_layers, None) if split is None else split
        if carry is None:
            dx = dout
            if c.maps is not None:
                c2p = c.maps.c2p
                if c.unpad is not None:                          # compact row -> row of the caller's unpadded layout
                    c2p = torch.where(c2p >= 0, self._pad_rows(c)[c2p.clamp(min=0)], c2p)
                dx = kn.rows_gather(dout, self.empty(c.maps.rows, cfg.d_model), c2p, live=c.maps.count)
            elif c.unpad is not None:
                dx = kn.rows_gather(dout, self.empty(c.Bn * c.S, cfg.d_model), self._pad_rows(c))
        else:
            dx = carry
        for i in reversed(range(lo, hi)):
            lc, fc = c.layers[i]
            dx = self._ffn_block_bwd(b + "layers.%d." % i, fc, dx)
            dx = self._self_block_bwd(b + "layers.%d." % i, lc, dx)
        if lo > 0:
            return dx
        if c.maps is not None:
            dx = kn.rows_gather(dx, self.empty(c.Bn * c.S, cfg.d_model), c.maps.p2c)
        kn.embed_ln_bwd(dx, c.ids, a.w(self.bp + "model.shared.weight"), a.w(b + "embed_positions.weight"), None, None,
                        a.f32(b + "layernorm_embedding.weight"), c.mean0, c.rstd0, a.g(self.bp + "model.shared.weight"),
                        a.g(b + "embed_positions.weight"), None, a.g(b + "layernorm_embedding.weight"),
                        a.g(b + "layernorm_embedding.bias"), c.Bn, c.S, cfg.extra_pos_embeddings, cfg.pad_token_id, c.p, c.seed0,
                        salt=self.salt)
        self.touch(self.bp + "model.shared.weight", b + "embed_positions.weight", b + "layernorm_embedding.weight",
                   b + "layernorm_embedding.bias")
        return None

    # ---- shared blocks ----------------------------------------------------------------------------
    def _self_block_fwd(self, lb, x, pad, Bn, T, causal, maps=None):
        """x -> LN(x + drop(out_proj(self_attention(x))))   (:288-297 / :442-461).
        maps (padding-free encoder): x holds the valid rows first (compact layout, live count on the device).  bf16: the
        attention kernels read q/k/v and write their output in that layout through the row map; f32: q/k/v are expanded to the
        padded [Bn*T, 3D] layout (zeros at padding) and the output is compacted again."""
        cfg, a = self.cfg, self.arena
        D, H = cfg.d_model, cfg.heads
        R = x.shape[0]
        live = maps.count if maps is not None else None
        q, k, v = self._attn_names(lb, "self_attn")
        c = NS(x=x, pad=pad, Bn=Bn, T=T, causal=causal, p=self.p_drop(), seed=self.next_seed(), maps=maps)
        c.qkv = self.empty(R, 3 * D)
        kn.gemm(x, a.wspan(q + ".weight", v + ".weight", (3 * D, D)), c.qkv, bias=a.span(a.data, q + ".bias", v + ".bias", (3 * D,)),
                live=live)
        c.mapped = maps is not None and self.dtype == torch.bfloat16      # the bf16 kernels read / write the compact layout through the row map
        if maps is not None and not c.mapped:
            c.qkv = kn.rows_gather(c.qkv, self.empty(Bn * T, 3 * D), maps.p2c)
        rmap = maps.p2c32 if c.mapped else None
        attn = self.empty(R if c.mapped else Bn * T, D)
        nq = 1
        c.long = None
        if T > 128 and causal:
            # a decoder sequence of 129 .. 224 positions (a training step on test.py's 158-token reviews): its first 128 queries as
            # the usual causal self-attention over the first 128 keys + the remaining ones as a second query block whose first row
            # sits at key 128 (mmsum_attn_desc.causal_q0) over all the keys
            assert maps is None
            attn = self._causal_long_fwd(c, pad, Bn, T)
        else:
            if T > 128:
                # more than 128 tokens per sequence (the encoder at test.py's 158-token reviews; encoder_fwd made T a multiple of the
                # split): `nq` query blocks of T / nq rows attend the sequence's T keys -- the descriptor of one entity shared by the
                # query blocks of a "business" (the table / image memory's form), forward and backward
                nq = self.seq_splits(T)
                assert T % nq == 0
            c.desc = kn.make_attn_desc(c.qkv[:, :D], c.qkv[:, D:2 * D], c.qkv[:, 2 * D:], attn, pad, None, Bn * nq, T // nq, nq, 1, T, H,
                                       False, causal, 64 ** -0.5, q_rows=rmap, kv_rows=rmap)
            kn.attn_fwd(c.desc, x)
        c.attn = attn if (maps is None or c.mapped) else kn.rows_gather(attn, self.empty(R, D), maps.c2p, live=live)
        c.o = self.empty(R, D)
        kn.gemm(c.attn, a.w(lb + "self_attn.out_proj.weight"), c.o, bias=a.f32(lb + "self_attn.out_proj.bias"), live=live)
        y = self.empty(R, D)
        c.mean, c.rstd = self.empty(R, dtype=torch.float32), self.empty(R, dtype=torch.float32)
        kn.add_ln_fwd(c.o, x, a.f32(lb + "self_attn_layer_norm.weight"), a.f32(lb + "self_attn_layer_norm.bias"), y, c.mean,
                      c.rstd, 1e-5, c.p, c.seed, salt=self.salt, live=live)
        return y, c

    def _self_block_bwd(self, lb, c, dy):
        cfg, a = self.cfg, self.arena
        D = cfg.d_model
        R = dy.shape[0]
        live = c.maps.count if c.maps is not None else None
        q, k, v = self._attn_names(lb, "self_attn")
        do, dx = self.empty(R, D), self.empty(R, D)
        kn.add_ln_bwd(dy, c.o, c.x, a.f32(lb + "self_attn_layer_norm.weight"), c.mean, c.rstd, do, dx, False,
                      a.g(lb + "self_attn_layer_norm.weight"), a.g(lb + "self_attn_layer_norm.bias"), c.p, c.seed,
                      dxsum=a.g(lb + "self_attn.out_proj.bias"), salt=self.salt, live=live)   # out_proj's bias gradient = column sums of do
        self.wgrad(do, c.attn, lb + "self_attn.out_proj.weight", live=live)
        dattn = self.empty(R, D)
        self.dgrad(do, lb + "self_attn.out_proj.weight", a.w(lb + "self_attn.out_proj.weight"), dattn, live=live)
        Rp = c.Bn * c.T
        if c.maps is not None and not c.mapped:
            dattn = kn.rows_gather(dattn, self.empty(Rp, D), c.maps.p2c)
        if c.long is not None:
            dqkv = self._causal_long_bwd(c, dattn)
        else:
            dqkv = self.empty(R if c.mapped else Rp, 3 * D)
            stats = self.empty(kn.attn_bwd_workspace(c.desc) // 4, dtype=torch.float32)
            kn.attn_bwd(c.desc, dattn, dqkv[:, :D], False, dqkv[:, D:2 * D], dqkv[:, 2 * D:], stats)
        if c.maps is not None and not c.mapped:
            dqkv = kn.rows_gather(dqkv, self.empty(R, 3 * D), c.maps.c2p, live=live)
        self.wgrad(dqkv, c.x, gview=a.gspan(q + ".weight", v + ".weight", (3 * D, D)), bias_g=a.gspan(q + ".bias", v + ".bias", (3 * D,)),
                   live=live)
        self.dgrad(dqkv, q + ".weight", a.wspan(q + ".weight", v + ".weight", (3 * D, D)), dx, accumulate=True, live=live)
        self.touch(q + ".weight", k + ".weight", v + ".weight", q + ".bias", k + ".bias", v + ".bias",
                   lb + "self_attn.out_proj.weight", lb + "self_attn.out_proj.bias", lb + "self_attn_layer_norm.weight",
                   lb + "self_attn_layer_norm.bias")
        return dx

    LONG_HEAD = 128          # queries of a long decoder sequence that run as the ordinary (<= 128-row) query block

    def _rows_of(self, x, Bn, T, lo, hi):
        """Rows lo .. hi-1 of every sequence of x [Bn*T, D] as a matrix of their own [Bn*(hi-lo), D]."""
        y = self.empty(Bn * (hi - lo), x.shape[1], dtype=x.dtype)
        y.view(Bn, hi - lo, x.shape[1]).copy_(x.view(Bn, T, x.shape[1])[:, lo:hi])
        return y

    def _causal_long_fwd(self, c, pad, Bn, T):
        """Causal self-attention over T in 129 .. 224 positions from c.qkv [Bn*T, 3D] -> [Bn*T, D] (see _self_block_fwd)."""
        D, H, T0 = self.cfg.d_model, self.cfg.heads, self.LONG_HEAD
        if T > 224:
            raise ValueError("decoder sequences of %d > 224 positions are not built (the attention kernels stage at most 224 keys)" % T)
        T1 = T - T0
        qkv3 = c.qkv.view(Bn, T, 3 * D)
        L = NS(T1=T1)
        L.qkv0 = self.empty(Bn * T0, 3 * D)
        L.qkv0.view(Bn, T0, 3 * D).copy_(qkv3[:, :T0])
        L.q1 = self.empty(Bn * T1, D)
        L.q1.view(Bn, T1, D).copy_(qkv3[:, T0:, :D])
        L.pad0 = None
        if pad is not None:
            L.pad0 = self.empty(Bn * T0, dtype=torch.uint8)
            L.pad0.view(Bn, T0).copy_(pad.reshape(Bn, T)[:, :T0])
        attn0, attn1 = self.empty(Bn * T0, D), self.empty(Bn * T1, D)
        L.d0 = kn.make_attn_desc(L.qkv0[:, :D], L.qkv0[:, D:2 * D], L.qkv0[:, 2 * D:], attn0, L.pad0, None, Bn, T0, 1, 1, T0, H, False, True,
                                 64 ** -0.5)
        L.d1 = kn.make_attn_desc(L.q1, c.qkv[:, D:2 * D], c.qkv[:, 2 * D:], attn1, pad, None, Bn, T1, 1, 1, T, H, False, True, 64 ** -0.5,
                                 causal_q0=T0)
        kn.attn_fwd(L.d0, c.qkv)
        kn.attn_fwd(L.d1, c.qkv)
        attn = self.empty(Bn * T, D)
        a3 = attn.view(Bn, T, D)
        a3[:, :T0].copy_(attn0.view(Bn, T0, D))
        a3[:, T0:].copy_(attn1.view(Bn, T1, D))
        c.long = L
        return attn

    def _causal_long_bwd(self, c, dattn):
        """dattn [Bn*T, D] -> dqkv [Bn*T, 3D]: the two query blocks' gradients; the second block's dK / dV cover every key."""
        D, T0 = self.cfg.d_model, self.LONG_HEAD
        L, Bn, T = c.long, c.Bn, c.T
        T1 = L.T1
        da3 = dattn.view(Bn, T, D)
        da0, da1 = self.empty(Bn * T0, D), self.empty(Bn * T1, D)
        da0.view(Bn, T0, D).copy_(da3[:, :T0])
        da1.view(Bn, T1, D).copy_(da3[:, T0:])
        dqkv0 = self.empty(Bn * T0, 3 * D)
        st0 = self.empty(kn.attn_bwd_workspace(L.d0) // 4, dtype=torch.float32)
        kn.attn_bwd(L.d0, da0, dqkv0[:, :D], False, dqkv0[:, D:2 * D], dqkv0[:, 2 * D:], st0)
        dq1, dkv1 = self.empty(Bn * T1, D), self.empty(Bn * T, 2 * D)
        st1 = self.empty(kn.attn_bwd_workspace(L.d1) // 4, dtype=torch.float32)
        kn.attn_bwd(L.d1, da1, dq1, False, dkv1[:, :D], dkv1[:, D:], st1)
        dqkv = self.empty(Bn * T, 3 * D)
        d3 = dqkv.view(Bn, T, 3 * D)
        d3[:, :T0].copy_(dqkv0.view(Bn, T0, 3 * D))
        d3[:, T0:, :D].copy_(dq1.view(Bn, T1, D))
        d3[:, T0:, D:].zero_()
        d3[:, :, D:].add_(dkv1.view(Bn, T, 2 * D))
        return dqkv

    def _ffn_block_fwd(self, lb, x, out=None, maps=None):
        """x -> LN(x + drop(fc2(gelu(fc1(x)))))   (:299-308 / :479-489)."""
        cfg, a = self.cfg, self.arena
        R, D = x.shape
        live = maps.count if maps is not None else None
        Fd = a.shapes[lb + "fc1.weight"][0]
        c = NS(x=x, p=self.p_drop(), seed=self.next_seed(), live=live)
        c.u, c.h = self.empty(R, Fd), self.empty(R, Fd)
        kn.gemm(x, a.w(lb + "fc1.weight"), c.h, bias=a.f32(lb + "fc1.bias"), epi=kn.EPI_GELU, aux=c.u, live=live)
        c.f = self.empty(R, D)
        kn.gemm(c.h, a.w(lb + "fc2.weight"), c.f, bias=a.f32(lb + "fc2.bias"), live=live)
        y = out if out is not None else self.empty(R, D)
        c.mean, c.rstd = self.empty(R, dtype=torch.float32), self.empty(R, dtype=torch.float32)
        kn.add_ln_fwd(c.f, x, a.f32(lb + "final_layer_norm.weight"), a.f32(lb + "final_layer_norm.bias"), y, c.mean, c.rstd,
                      1e-5, c.p, c.seed, salt=self.salt, live=live)
        return y, c

    def _ffn_block_bwd(self, lb, c, dy):
        a = self.arena
        R, D = dy.shape
        live = c.live
        df, dx = self.empty(R, D), self.empty(R, D)
        kn.add_ln_bwd(dy, c.f, c.x, a.f32(lb + "final_layer_norm.weight"), c.mean, c.rstd, df, dx, False,
                      a.g(lb + "final_layer_norm.weight"), a.g(lb + "final_layer_norm.bias"), c.p, c.seed,
                      dxsum=a.g(lb + "fc2.bias"), salt=self.salt, live=live)                 # fc2's bias gradient = column sums of df
        self.wgrad(df, c.h, lb + "fc2.weight", live=live)
        du = self.empty(R, c.u.shape[1])
        self.dgrad(df, lb + "fc2.weight", a.w(lb + "fc2.weight"), du, epi=kn.EPI_GELU_BWD, aux=c.u, colsum=a.g(lb + "fc1.bias"), live=live)
        self.wgrad(du, c.x, lb + "fc1.weight", live=live)
        self.dgrad(du, lb + "fc1.weight", a.w(lb + "fc1.weight"), dx, accumulate=True, live=live)
        self.touch(lb + "fc1.weight", lb + "fc1.bias", lb + "fc2.weight", lb + "fc2.bias", lb + "final_layer_norm.weight",
                   lb + "final_layer_norm.bias")
        return dx

    # =============================================================================================
    # Decoder  (BartDecoder.forward :530-660, DecoderLayer :432-494, SelfAttention.forward :711-750)
    # =============================================================================================
    def make_memory(self, B, mods):
        """mods: list of (N, S) per modality.  Returns the layout of the concatenated memory
        matrix [Rmem, D]: rows of modality m start at off[m]; entity (b,n) at off[m] + (b*N+n)*S."""
        offs, off = [], 0
        for N, S in mods:
            offs.append(off)
            off += B * N * S
        return NS(B=B, mods=list(mods), offs=offs, rows=off)

    def decoder_fwd(self, dec_ids, dec_pad, rating_diff, mem, layout, pads, qpb, exclude_self, compact_mem=False):
        """dec_ids [Bd,T]; dec_pad [Bd,T] uint8 or None; rating_diff [Bd] f32 or None; mem [Rmem,D];
        pads: per-modality uint8 [B,N,S] (1 = padded key).  Bd = B*qpb.
        compact_mem (fused step): the cross-attention K/V projections (and their gradients) run on the memory rows that
        are not masked keys only (live count on the device); K/V are expanded to the padded layout the attention kernel
        reads (masked rows zero) once per layer."""
        cfg, a = self.cfg, self.arena
        D, H = cfg.d_model, cfg.heads
        Bd, T = dec_ids.shape
        Rq = Bd * T
        b = self.bp + "model.decoder."
        nm = len(layout.mods)
        c = NS(Bd=Bd, T=T, ids=dec_ids.contiguous(), rd=rating_diff, mem=mem, layout=layout, pads=pads, qpb=qpb,
               exclude_self=exclude_self, layers=[], p=self.p_drop(), seed0=self.next_seed(), dec_pad=dec_pad, mem_maps=None, mem_c=mem)
        if compact_mem:
            keep = torch.cat([pd.reshape(-1) for pd in pads]).eq(0)            # rows that are real keys, memory-row order
            c.mem_maps = self.row_maps(keep)
            c.mem_c = kn.rows_gather(mem, self.empty(layout.rows, D), c.mem_maps.c2p, live=c.mem_maps.count)
        # null-entity flags per modality (:858) and the per-business no-table / no-image flags (:732-736)
        c.nulls = []
        for (N, S), pad in zip(layout.mods, pads):
            nul = self.empty(layout.B * N, dtype=torch.uint8)
            kn.entity_null(pad, nul, layout.B * N, S)
            c.nulls.append(nul)
        if self.multimodal:
            c.no_table = c.nulls[1]
            N2, S2 = layout.mods[2]
            c.no_img = self.empty(layout.B, dtype=torch.uint8)
            kn.entity_null(pads[2], c.no_img, layout.B, N2 * S2)
        x = self.empty(Rq, D)
        c.mean0, c.rstd0 = self.empty(Rq, dtype=torch.float32), self.empty(Rq, dtype=torch.float32)
        kn.embed_ln_fwd(c.ids, a.w(self.bp + "model.shared.weight"), a.w(b + "embed_positions.weight"), rating_diff,
                        a.w(b + "rating_embeddings") if rating_diff is not None else None, a.f32(b + "layernorm_embedding.weight"),
                        a.f32(b + "layernorm_embedding.bias"), x, c.mean0, c.rstd0, Bd, T, cfg.extra_pos_embeddings, 1e-5, c.p,
                        c.seed0, salt=self.salt)
        for i in range(cfg.decoder_layers):
            lb = b + "layers.%d." % i
            x, sc = self._self_block_fwd(lb, x, dec_pad, Bd, T, causal=True)
            x, cc = self._cross_block_fwd(lb, x, c)
            x, fc = self._ffn_block_fwd(lb, x)
            c.layers.append((sc, cc, fc))
        c.out = x
        return x, c

    def decoder_bwd(self, c, dout, split=None):
        """dout [Rq, D] (consumed) -> dmem [Rmem, D]; accumulates decoder parameter gradients.
        split=(lo, hi, carry): run layers hi-1 .. lo only -- the fused step cuts the decoder backward into gradient segments of a
        few layers each, so that the data-parallel exchange of a finished segment runs under the next one.  `carry` = (dx, dmem)
        handed from the part above (None for the part that starts at the top); a part with lo > 0 returns it for the next."""
        cfg, a = self.cfg, self.arena
        b = self.bp + "model.decoder."
        L = cfg.decoder_layers
        lo, hi, carry = (0, L, None) if split is None else split
        if carry is None:
            dmem = self.empty(c.mem_c.shape[0], cfg.d_model)          # compact rows when the K/V projections run padding-free
            dx = dout
        else:
            dx, dmem = carry
        for i in reversed(range(lo, hi)):
            lb = b + "layers.%d." % i
            sc, cc, fc = c.layers[i]
            dx = self._ffn_block_bwd(lb, fc, dx)
            dx = self._cross_block_bwd(lb, cc, c, dx, dmem, first=(i == L - 1))
            dx = self._self_block_bwd(lb, sc, dx)
        if lo > 0:
            return dx, dmem
        has_r = c.rd is not None
        kn.embed_ln_bwd(dx, c.ids, a.w(self.bp + "model.shared.weight"), a.w(b + "embed_positions.weight"), c.rd,
                        a.w(b + "rating_embeddings") if has_r else None, a.f32(b + "layernorm_embedding.weight"), c.mean0, c.rstd0,
                        a.g(self.bp + "model.shared.weight"), a.g(b + "embed_positions.weight"),
                        a.g(b + "rating_embeddings") if has_r else None, a.g(b + "layernorm_embedding.weight"),
                        a.g(b + "layernorm_embedding.bias"), c.Bd, c.T, cfg.extra_pos_embeddings, cfg.pad_token_id, c.p, c.seed0,
                        salt=self.salt)
        self.touch(self.bp + "model.shared.weight", b + "embed_positions.weight", b + "layernorm_embedding.weight",
                   b + "layernorm_embedding.bias")
        if has_r:
            self.touch(b + "rating_embeddings")
        if c.mem_maps is not None:
            dmem = kn.rows_gather(dmem, self.empty(c.layout.rows, cfg.d_model), c.mem_maps.p2c)
        return dmem

    def _cross_block_fwd(self, lb, x, dc):
        cfg, a = self.cfg, self.arena
        D, H = cfg.d_model, cfg.heads
        Rq = x.shape[0]
        L = dc.layout
        nm = len(L.mods)
        q, k, v = self._attn_names(lb, "encoder_attn")
        pre = lb + "encoder_attn."
        c = NS(x=x, p=self.p_drop(), seed=self.next_seed())
        c.q = self.empty(Rq, D)
        kn.gemm(x, a.w(q + ".weight"), c.q, bias=a.f32(q + ".bias"))                                     # :783 (scale folded into the kernel)
        mlive = dc.mem_maps.count if dc.mem_maps is not None else None
        c.kv = self.empty(dc.mem_c.shape[0], 2 * D)
        kn.gemm(dc.mem_c, a.wspan(k + ".weight", v + ".weight", (2 * D, D)), c.kv,
                bias=a.span(a.data, k + ".bias", v + ".bias", (2 * D,)), live=mlive)                      # :788-789, hoisted
        # bf16: K / V stay compact, read through the row map (not for the two query blocks of a long decoder sequence: their dK / dV are
        # added per modality slice of the padded layout)
        c.mapped = dc.mem_maps is not None and self.dtype == torch.bfloat16 and dc.T <= 128
        if dc.mem_maps is not None and not c.mapped:
            c.kv = kn.rows_gather(c.kv, self.empty(L.rows, 2 * D), dc.mem_maps.p2c)
        c.heads = self.empty(nm * Rq, D)
        c.descs = []
        # query blocks hold at most 128 rows: a decoder sequence of 129 .. 224 positions runs as two query blocks per (business, pass)
        # -- its first 128 rows and the rest, each gathered into a matrix of its own -- over the same memory
        parts = [(0, dc.T)] if dc.T <= 128 else [(0, self.LONG_HEAD), (self.LONG_HEAD, dc.T)]
        if dc.T > 224:
            raise ValueError("decoder sequences of %d > 224 positions are not built" % dc.T)
        c.parts = parts
        c.qparts = [c.q] if len(parts) == 1 else [self._rows_of(c.q, dc.Bd, dc.T, lo, hi) for lo, hi in parts]
        for m, ((N, S), pad) in enumerate(zip(L.mods, dc.pads)):
            rows = slice(L.offs[m], L.offs[m] + L.B * N * S)
            heads_m = c.heads[m * Rq:(m + 1) * Rq]
            ds = []
            for (lo, hi), qp in zip(parts, c.qparts):
                out = heads_m if len(parts) == 1 else self.empty(dc.Bd * (hi - lo), D)
                if c.mapped:  # physical rows are rows of the whole compact matrix; the map is the modality's slice of padded -> compact
                    d = kn.make_attn_desc(qp, c.kv[:, :D], c.kv[:, D:], out, pad, dc.nulls[m], dc.Bd, hi - lo, dc.qpb, N, S,
                                          H, dc.exclude_self and m == 0, False, 64 ** -0.5, kv_rows=dc.mem_maps.p2c32[rows])
                else:
                    d = kn.make_attn_desc(qp, c.kv[rows, :D], c.kv[rows, D:], out, pad, dc.nulls[m],
                                          dc.Bd, hi - lo, dc.qpb, N, S, H, dc.exclude_self and m == 0, False, 64 ** -0.5)
                kn.attn_fwd(d, x)                                                                          # :819-869
                if len(parts) > 1:
                    heads_m.view(dc.Bd, dc.T, D)[:, lo:hi].copy_(out.view(dc.Bd, hi - lo, D))
                ds.append(d)
            c.descs.append(ds)
        c.y = self.empty(nm * Rq, D)
        kn.gemm(c.heads, a.w(pre + "out_proj.weight"), c.y, bias=a.f32(pre + "out_proj.bias"))             # :728-730 / :885
        if self.multimodal:
            yt, ytab, yimg = c.y[:Rq], c.y[Rq:2 * Rq], c.y[2 * Rq:]
            c.pa, c.pb = self.empty(Rq, D), self.empty(Rq, D)
            kn.gemm(yt, a.w(pre + "alpha_proj.weight"), c.pa, a2=ytab, bias=a.f32(pre + "alpha_proj.bias"))  # :738
            kn.gemm(yt, a.w(pre + "beta_proj.weight"), c.pb, a2=yimg, bias=a.f32(pre + "beta_proj.bias"))    # :739
            c.c = self.empty(Rq, D)
            kn.gate_fwd(c.pa, c.pb, yt, ytab, yimg, dc.no_table, dc.no_img, c.c, dc.qpb * dc.T)              # :740-744
        else:
            c.c = c.y
        y = self.empty(Rq, D)
        c.mean, c.rstd = self.empty(Rq, dtype=torch.float32), self.empty(Rq, dtype=torch.float32)
        kn.add_ln_fwd(c.c, x, a.f32(lb + "encoder_attn_layer_norm.weight"), a.f32(lb + "encoder_attn_layer_norm.bias"), y, c.mean,
                      c.rstd, 1e-5, c.p, c.seed, salt=self.salt)
        return y, c

    def _cross_block_bwd(self, lb, c, dc, dy, dmem, first):
        cfg, a = self.cfg, self.arena
        D = cfg.d_model
        Rq = dy.shape[0]
        L = dc.layout
        nm = len(L.mods)
        q, k, v = self._attn_names(lb, "encoder_attn")
        pre = lb + "encoder_attn."
        dcv, dx = self.empty(Rq, D), self.empty(Rq, D)
        kn.add_ln_bwd(dy, c.c, c.x, a.f32(lb + "encoder_attn_layer_norm.weight"), c.mean, c.rstd, dcv, dx, False,
                      a.g(lb + "encoder_attn_layer_norm.weight"), a.g(lb + "encoder_attn_layer_norm.bias"), c.p, c.seed, salt=self.salt)
        if self.multimodal:
            yt, ytab, yimg = c.y[:Rq], c.y[Rq:2 * Rq], c.y[2 * Rq:]
            dyy = self.empty(3 * Rq, D)
            dyt, dytab, dyimg = dyy[:Rq], dyy[Rq:2 * Rq], dyy[2 * Rq:]
            dpa, dpb = self.empty(Rq, D), self.empty(Rq, D)
            # the alpha / beta bias gradients (column sums of dpa / dpb) are taken inside the gate kernel (f32 atomics); the
            # deterministic mode keeps the separate, order-fixed column-sum passes
            fused_b = not self.deterministic
            sums = (a.g(pre + "alpha_proj.bias"), a.g(pre + "beta_proj.bias")) if fused_b else None
            kn.gate_bwd(dcv, c.pa, c.pb, ytab, yimg, dc.no_table, dc.no_img, dpa, dpb, dyt, dytab, dyimg, dc.qpb * dc.T, sums=sums)
            for name, dp, other, dother in ((pre + "alpha_proj", dpa, ytab, dytab), (pre + "beta_proj", dpb, yimg, dyimg)):
                W, gW = a.w(name + ".weight"), a.g(name + ".weight")
                if not fused_b:
                    self.bgrad(dp, name + ".bias")
                self.wgrad(dp, yt, gview=gW[:, :D])
                self.wgrad(dp, other, gview=gW[:, D:])
                self.dgrad(dp, name + ".weight", W[:, :D], dyt, accumulate=True, rows=slice(0, D))
                self.dgrad(dp, name + ".weight", W[:, D:], dother, accumulate=True, rows=slice(D, 2 * D))
                self.touch(name + ".weight", name + ".bias")
        else:
            dyy = dcv
        self.wgrad(dyy, c.heads, pre + "out_proj.weight", bias_g=a.g(pre + "out_proj.bias"))
        dheads = self.empty(nm * Rq, D)
        self.dgrad(dyy, pre + "out_proj.weight", a.w(pre + "out_proj.weight"), dheads)
        dq = self.empty(Rq, D)
        dkv = self.empty(dc.mem_maps.rows if c.mapped else L.rows, 2 * D)
        long = len(c.parts) > 1
        dqs = [dq] if not long else [self.empty(dc.Bd * (hi - lo), D) for lo, hi in c.parts]
        for m, (N, S) in enumerate(L.mods):
            rows = slice(None) if c.mapped else slice(L.offs[m], L.offs[m] + L.B * N * S)
            dh_m = dheads[m * Rq:(m + 1) * Rq]
            for pi, ((lo, hi), d) in enumerate(zip(c.parts, c.descs[m])):
                dh = dh_m if not long else self._rows_of(dh_m, dc.Bd, dc.T, lo, hi)
                # every launch overwrites the dK / dV rows of its modality: the second query block's go to a buffer of their own and are added
                dkv_p = dkv if pi == 0 else self.empty(dkv.shape[0], 2 * D)
                stats = self.empty(kn.attn_bwd_workspace(d) // 4, dtype=torch.float32)
                kn.attn_bwd(d, dh, dqs[pi], m > 0, dkv_p[rows, :D], dkv_p[rows, D:], stats)
                if pi > 0:
                    dkv[rows].add_(dkv_p[rows])
        if long:
            for (lo, hi), dqp in zip(c.parts, dqs):
                dq.view(dc.Bd, dc.T, D)[:, lo:hi].copy_(dqp.view(dc.Bd, hi - lo, D))
        mlive = None
        if dc.mem_maps is not None:
            mlive = dc.mem_maps.count
            if not c.mapped:
                dkv = kn.rows_gather(dkv, self.empty(dc.mem_maps.rows, 2 * D), dc.mem_maps.c2p, live=mlive)
        self.wgrad(dkv, dc.mem_c, gview=a.gspan(k + ".weight", v + ".weight", (2 * D, D)), bias_g=a.gspan(k + ".bias", v + ".bias", (2 * D,)),
                   live=mlive)
        self.dgrad(dkv, k + ".weight", a.wspan(k + ".weight", v + ".weight", (2 * D, D)), dmem, accumulate=not first, live=mlive)
        self.wgrad(dq, c.x, q + ".weight", bias_g=a.g(q + ".bias"))
        self.dgrad(dq, q + ".weight", a.w(q + ".weight"), dx, accumulate=True)
        self.touch(q + ".weight", k + ".weight", v + ".weight", q + ".bias", k + ".bias", v + ".bias", pre + "out_proj.weight",
                   pre + "out_proj.bias", lb + "encoder_attn_layer_norm.weight", lb + "encoder_attn_layer_norm.bias")
        return dx

    # =============================================================================================
    # LM head (+ fused label-smoothing loss)   (:2281; utils.py:32-38)
    # =============================================================================================
    def lm_logits_fwd(self, h):
        """-> logits buffer [Rq, Vpad] (columns >= V are scratch)."""
        logits = self.empty(h.shape[0], self.Vpad)
        kn.gemm(h, self.arena.w(self.bp + "model.shared.weight"), logits[:, :self.cfg.vocab_size])
        return logits

    def lm_head_bwd(self, h, dlogits, upstream=None):
        """dlogits [Rq, Vpad] with zero padding columns.  -> dh; accumulates the tied-embedding gradient.
        upstream: device f32 scalar, the gradient arriving at the loss (loss.backward(g), loss / accumulation_steps):
        every gradient of the step is linear in dlogits, so scaling the two products that read it scales them all."""
        V, name = self.cfg.vocab_size, self.bp + "model.shared.weight"
        dh = self.empty(h.shape[0], h.shape[1])
        if name in self.wt:
            kn.gemm(dlogits, self.wt[name], dh, alpha_dev=upstream)          # K = Vpad: padding columns of both operands are zero
        else:
            kn.gemm(dlogits[:, :V], self.arena.w(name), dh, b_t=True, alpha_dev=upstream)
        self.wgrad(dlogits[:, :V], h, name, alpha_dev=upstream)
        self.touch(name)
        return dh

    def lm_loss_fwd(self, h, labels, smoothing, n_segments):
        """Fused LM head + loss.  Returns (mean loss [1], per-segment losses [n_segments], dlogits)."""
        Rq, V = h.shape[0], self.cfg.vocab_size
        logits = self.lm_logits_fwd(h)
        rows = self.empty(Rq, dtype=torch.float32)
        kn.ls_loss(logits, labels.reshape(-1).contiguous(), rows, V, float(smoothing or 0.0), 1.0 / Rq, True)
        seg = self.empty(n_segments, dtype=torch.float32)       # per decoder sequence (b, i): mean over its T rows
        kn.segment_sum(rows, seg, n_segments, Rq // n_segments, float(n_segments) / Rq)
        loss = self.empty(1, dtype=torch.float32)
        kn.segment_sum(rows, loss, 1, Rq, 1.0 / Rq)
        return loss, seg, logits

    # =============================================================================================
    # Table encoder  (table_encoder.py:14-83)
    # =============================================================================================
    def table_fwd(self, field, fv, out=None):
        a = self.arena
        B = fv[0].shape[0]
        D = self.cfg.d_model
        P = self.table_positions
        tp = "table_encoder."
        c = NS(B=B, fv=[t.contiguous() for t in fv])
        c.all = self.empty(B * P, 2 * D)
        c.mask = self.empty(B, P, dtype=torch.uint8)
        if self.table_kind == "yelp":
            kn.table_gather(a.w(self.bp + "model.shared.weight"), field.contiguous(), c.fv, a.w(tp + "rating_embedding.weight"),
                            a.w(tp + "hours_embedding.weight"), c.all, c.mask, B, self.cfg.pad_token_id)
        else:
            kn.amazon_table_gather(a.w(self.bp + "model.shared.weight"), field.contiguous(), c.fv, a.w(tp + "price_embedding.weight"),
                                   a.w(tp + "rating_embedding.weight"), c.all, c.mask, B, self.cfg.pad_token_id)
        c.t1 = self.empty(B * P, D)
        kn.gemm(c.all, a.w(tp + "fc.weight"), c.t1, bias=a.f32(tp + "fc.bias"), epi=kn.EPI_RELU)
        y = out if out is not None else self.empty(B * P, D)
        kn.gemm(c.t1, a.w(tp + "linear.weight"), y)
        return y, c

    def table_bwd(self, c, dy):
        a = self.arena
        D = self.cfg.d_model
        P = self.table_positions
        tp = "table_encoder."
        self.wgrad(dy, c.t1, tp + "linear.weight")
        dt1 = self.empty(c.B * P, D)
        self.dgrad(dy, tp + "linear.weight", a.w(tp + "linear.weight"), dt1, epi=kn.EPI_RELU_BWD, aux=c.t1)
        self.wgrad(dt1, c.all, tp + "fc.weight", bias_g=a.g(tp + "fc.bias"))
        dall = self.empty(c.B * P, 2 * D)
        self.dgrad(dt1, tp + "fc.weight", a.w(tp + "fc.weight"), dall)
        if self.table_kind == "yelp":
            kn.table_gather_bwd(dall, c.fv[4], c.fv[5], a.g(tp + "rating_embedding.weight"), a.g(tp + "hours_embedding.weight"), c.B, D)
            self.touch(tp + "rating_embedding.weight", tp + "hours_embedding.weight")
        else:
            kn.amazon_table_gather_bwd(dall, c.fv[0], c.fv[1], a.g(tp + "price_embedding.weight"), a.g(tp + "rating_embedding.weight"), c.B, D)
            self.touch(tp + "price_embedding.weight", tp + "rating_embedding.weight")
        self.touch(tp + "linear.weight", tp + "fc.bias", tp + "fc.weight")

    # =============================================================================================
    # ResNet101 stages 1-3 + projection  (img_encoder.py:31-41; torchvision 0.6.1 resnet101)
    # =============================================================================================
    def _build_conv_mats(self, all_layers):
        """KxK conv weights as [Cout, Kpad] matrices with (kh,kw,c) column order, compute dtype."""
        a = self.arena
        r = "img_encoder.resnet."
        todo = []
        if all_layers:
            todo.append((r + "conv1.weight", 64, 3, 7))
        for li, bi, inp, pl, stride, down in resnet_blocks():
            if li <= 3 and (all_layers or li == 3):
                todo.append((r + "layer%d.%d.conv2.weight" % (li, bi), pl, pl, 3, stride))
        for name, co, ci, ks, *rest in todo:
            Kpad = (ks * ks * ci + 63) // 64 * 64
            m = self.conv_mats.get(name)
            if m is None:
                m = self.empty(co, Kpad)
                self.conv_mats[name] = m
            kn.conv_weight_to_matrix(m, a.f32(name), co, ci, ks, ks, Kpad)
            if self.dtype == torch.bfloat16 and ".layer3." in name:
                if rest[0] == 1 and self._implicit_bwd_ok(ci, co):
                    # stride 1: the input gradient is the forward's implicit convolution of the padded output gradient with the rotated weights
                    mr = self.conv_mats_r.get(name)
                    if mr is None:
                        mr = self.empty(ci, ks * ks * co)
                        self.conv_mats_r[name] = mr
                    kn.conv_weight_to_dgrad_matrix(mr, a.f32(name), co, ci, ks, ks, ks * ks * co)
                    continue
                mt = self.conv_mats_t.get(name)
                if mt is None:
                    mt = self.empty(Kpad, co)
                    self.conv_mats_t[name] = mt
                kn.transpose(m, mt)

    def _implicit_bwd_ok(self, cin, cout):
        """The backward of a stride-1 3x3 convolution without im2col / col2im (kn.conv3x3_wgrad + the input gradient as a convolution with the
        rotated weights): bf16 step, channel counts the four-wave TN kernel's tiles fit (one tap per 256-column tile)."""
        return (self.implicit_conv and os.environ.get("MMSUM_IMPLICIT_CONV") != "fwd" and self.dtype == torch.bfloat16 and cin >= 256 and (cin & (cin - 1)) == 0
                and cout >= 64 and (cout & (cout - 1)) == 0)

    def _bn_fwd(self, name, x, relu, residual=None, raw=None, pad_hw=None):
        """raw (f32 [2C], optional): {sum x, sum x^2} over the rows of x, left by the convolution's GEMM epilogue (_conv_gemm): the
        statistics then cost one tiny launch instead of a pass over x.
        pad_hw = (H, W): the output is written in the zero-bordered padded NHWC layout [n, H+2, W+2, C] -- the operand of the implicit
        3x3 convolution that follows (kn.conv3x3_gemm); the backward pass reads its ReLU mask from the same layout."""
        a = self.arena
        R, C = x.shape
        c = NS(x=x, relu=relu, name=name, pad_hw=pad_hw)
        if pad_hw is not None:
            H, W = pad_hw
            c.y = self.zeros(R // (H * W) * (H + 2) * (W + 2), C)      # the borders stay zero: the kernel writes the interior only
        else:
            c.y = self.empty(R, C)
        c.sums = self.empty(2 * C, dtype=torch.float32)
        training = self.training
        rm, rv = self.buffers[name + ".running_mean"], self.buffers[name + ".running_var"]
        ip = self._ip                                  # live-image window (None: every image runs)
        if training and raw is None:
            kn.bn_reduce(x, c.sums, images=ip)
        elif training and ip is not None:
            kn.bn_rep_fix(x, raw, ip)                  # the epilogue counted the representative's rows once: add the other (multiplicity - 1) shares
        # raw: the apply kernel derives the statistics from the epilogue's sums, writes c.sums and updates the running statistics itself
        kn.bn_apply(x, c.sums, a.f32(name + ".weight"), a.f32(name + ".bias"), residual, c.y, rm, rv, 1e-5, 0.1, relu, training, pad_hw=pad_hw,
                    raw=raw if training else None, images=ip)
        return c.y, c

    def _conv_gemm(self, x, w, bn_name=None):
        """y = x w^T for a convolution lowered to a GEMM (1x1: x = the NHWC activations; KxK: x = the im2col matrix).  In the bf16
        training step the BatchNorm statistics of y are taken in the GEMM's epilogue (column sums of y and y^2 of the values as
        stored, f32 atomics): returns (y, raw) with raw = the 2C sums for _bn_fwd, or None where the separate reduction runs."""
        y = self.empty(x.shape[0], w.shape[0])
        raw = self._bn_raw_slot(w.shape[0]) if (bn_name is not None and kn.gemm_colsum_fusable(x)) else None
        kn.gemm(x, w, y, colsum=raw, colsum_sq=raw is not None, live=self._lv(x.shape[0]))
        return y, raw

    def _bn_raw_slot(self, cout):
        """2 * cout zeroed floats of the step's statistics buffer for a convolution whose GEMM epilogue leaves {sum y, sum y^2} (bf16
        training step only; None otherwise: the separate reduction runs)."""
        if not (self.training and self.dtype == torch.bfloat16 and not self.deterministic and self._bn_raw is not None):
            return None
        n = 2 * cout
        raw = self._bn_raw[self._bn_raw_off:self._bn_raw_off + n]
        self._bn_raw_off += (n + 63) // 64 * 64
        assert self._bn_raw_off <= self._bn_raw.numel()
        return raw

    def _conv3x3_implicit(self, xp, w, n, H, W, C, bn_name):
        """3x3 / stride 1 / padding 1 convolution of the padded activations xp as an implicit GEMM (no im2col matrix): -> (y compact, raw)."""
        y = self.empty(n * H * W, w.shape[0])
        raw = self._bn_raw_slot(w.shape[0]) if bn_name is not None else None
        kn.conv3x3_gemm(xp, w, y, n, H, W, C, stats=raw, live=self._lv(n * H * W))
        return y, raw

    def _bn_bwd(self, c, dy, dresidual=None, dx_padded=None, dx_pad_hw=None):
        """dx_padded / dx_pad_hw: write dx into this zero-bordered padded buffer (interior only) instead of a compact matrix."""
        a = self.arena
        R, C = dy.shape
        dsums = self.empty(2 * C, dtype=torch.float32)
        pad_hw = getattr(c, "pad_hw", None)                  # the forward output (ReLU mask) sits in the padded layout
        kn.bn_bwd_reduce(dy, c.y, c.x, c.sums, dsums, 1e-5, c.relu, pad_hw=pad_hw, images=self._ip)
        dx = dx_padded if dx_padded is not None else self.empty(R, C)
        kn.bn_bwd_apply(dy, c.y, c.x, c.sums, dsums, a.f32(c.name + ".weight"), dx, dresidual, a.g(c.name + ".weight"),
                        a.g(c.name + ".bias"), 1e-5, c.relu, pad_hw=pad_hw, dx_pad_hw=dx_pad_hw, images=self._ip)
        self.touch(c.name + ".weight", c.name + ".bias")
        return dx

    def _conv1x1(self, x, name, bn_name=None):
        w = self.arena.w(name)
        return self._conv_gemm(x, w.view(w.shape[0], w.shape[1]), bn_name)

    def _lv(self, rows, adjust=0):
        """Device row count of a [rows, .] image-branch matrix under the live-image window (None: every row)."""
        ip = self._ip
        return None if ip is None else ip.rows(rows // ip.n, adjust)

    def _image_row_kinds(self, Hh, Ww):
        """(rows per image, adjustment) of every matrix of the image branch whose GEMM needs a device row count: the compact activations of
        each resolution, and the padded positions of layer3's implicit weight-gradient reduction (mmsum_conv3x3_wgrad)."""
        kinds = []

        def add(k):
            if k not in kinds:
                kinds.append(k)
        H, W = (Hh + 6 - 7) // 2 + 1, (Ww + 6 - 7) // 2 + 1
        add((H * W, 0))
        H, W = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        add((H * W, 0))
        for li, bi, inp, pl, stride, down in resnet_blocks():
            if li > 3:
                break
            if li == 3 and stride == 1:
                add(((H + 2) * (W + 2), -2 * (W + 3)))
            H, W = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
            add((H * W, 0))
        return kinds, H * W

    def img_fwd(self, img, out=None, img_mask=None):
        """img [n,3,H,W] f32 NCHW -> [n*196, D] (rows (n, h, w)); saves what layer3's backward needs.
        img_mask ([n], non-zero = a real image; the fused step passes it): the EMPTY slots -- masked and all zero, the padding
        data_utils.py:54-65 adds up to the batch's image count -- are identical inputs, so ONE representative runs for all of them with a
        multiplicity in the BatchNorm sums and (through its gradient rows) in the weight gradients; the filled slots run first, in batch
        order (kn.image_plan: device-side, one captured graph serves every batch).  Results are those of running every slot."""
        a = self.arena
        r = "img_encoder.resnet."
        n, _, Hh, Ww = img.shape
        c = NS(n=n, blocks=[])
        img = img.contiguous()
        ip = None
        if img_mask is not None and n <= 8192 and os.environ.get("MMSUM_IMAGE_DEDUPE") != "0":      # (mmsum_image_plan plans up to 8192 slots)
            kinds, positions = self._image_row_kinds(Hh, Ww)
            ip = kn.image_plan(img, img_mask, kn.ImagePlan(n, positions, kinds, img.device))
        self._ip = c.ip = ip
        x = self.empty(n * Hh * Ww, 3)
        kn.nchw_to_nhwc(img, x, n, 3, Hh, Ww, images=ip)
        if self.training:
            self._nbt_all[:self._nbt_live] += 1          # BatchNorm num_batches_tracked of every layer this pass runs
        # one zeroed buffer for the {sum, sum of squares} every convolution's epilogue accumulates (stages 1-3: 94 BatchNorm layers)
        self._bn_raw, self._bn_raw_off = None, 0
        if self.training and self.dtype == torch.bfloat16 and not self.deterministic:
            self._bn_raw = self.zeros(2 * 64 * 1024, dtype=torch.float32)
        Ho, Wo = (Hh + 6 - 7) // 2 + 1, (Ww + 6 - 7) // 2 + 1
        wm = self.conv_mats[r + "conv1.weight"]
        col = self.empty(n * Ho * Wo, wm.shape[1])
        kn.im2col(x, col, n, Hh, Ww, 3, 7, 7, 2, 3, Ho, Wo, wm.shape[1], images=ip)
        y, raw = self._conv_gemm(col, wm, r + "bn1")
        y, _ = self._bn_fwd(r + "bn1", y, True, raw=raw)
        Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
        x = self.empty(n * Hp * Wp, 64)
        kn.maxpool3x3s2(y, x, n, Ho, Wo, 64, Hp, Wp, images=ip)
        Hc, Wc = Hp, Wp
        for li, bi, inp, pl, stride, down in resnet_blocks():
            if li > 3:
                break
            b = r + "layer%d.%d." % (li, bi)
            bc = NS(x=x, H=Hc, W=Wc, inp=inp, pl=pl, stride=stride, down=down, name=b, li=li)
            c1, raw = self._conv1x1(x, b + "conv1.weight", b + "bn1")
            # stride-1 3x3 convolutions (every bottleneck but layer2.0 / layer3.0) run as IMPLICIT GEMMs in the bf16 step: bn1 writes its
            # output in the zero-bordered padded layout and the NT kernels' DMA pieces read one tap's channels straight from it
            implicit = self.implicit_conv and self.dtype == torch.bfloat16 and stride == 1 and pl >= 64 and (pl & (pl - 1)) == 0
            o1, bc.bn1 = self._bn_fwd(b + "bn1", c1, True, raw=raw, pad_hw=(Hc, Wc) if implicit else None)
            Ho2, Wo2 = (Hc + 2 - 3) // stride + 1, (Wc + 2 - 3) // stride + 1
            wm = self.conv_mats[b + "conv2.weight"]
            if implicit:
                bc.col = None                                  # layer3's weight gradient re-creates it from the padded o1 (img_bwd)
                c2, raw = self._conv3x3_implicit(o1, wm, n, Hc, Wc, pl, b + "bn2")
            else:
                bc.col = self.empty(n * Ho2 * Wo2, wm.shape[1])
                kn.im2col(o1, bc.col, n, Hc, Wc, pl, 3, 3, stride, 1, Ho2, Wo2, wm.shape[1], images=ip)
                c2, raw = self._conv_gemm(bc.col, wm, b + "bn2")
            o2, bc.bn2 = self._bn_fwd(b + "bn2", c2, True, raw=raw)
            bc.o1, bc.o2 = o1, o2
            c3, raw3 = self._conv1x1(o2, b + "conv3.weight", b + "bn3")
            if down:
                if stride == 1:
                    xs = x
                else:
                    xs = self.empty(n * Ho2 * Wo2, inp)
                    kn.im2col(x, xs, n, Hc, Wc, inp, 1, 1, stride, 0, Ho2, Wo2, inp, images=ip)
                bc.xs = xs
                cd, rawd = self._conv1x1(xs, b + "downsample.0.weight", b + "downsample.1")
                idt, bc.bnd = self._bn_fwd(b + "downsample.1", cd, False, raw=rawd)
            else:
                idt = x
            x, bc.bn3 = self._bn_fwd(b + "bn3", c3, True, residual=idt, raw=raw3)
            Hc, Wc = Ho2, Wo2
            if li == 3:
                c.blocks.append(bc)
        self._bn_raw = None
        c.feat = x                                   # [n*14*14, 1024] for 224x224 inputs
        y = out if out is not None else self.empty(x.shape[0], self.cfg.d_model)
        if ip is None:
            kn.gemm(x, a.w("img_encoder.linear.weight"), y)
        else:          # run order -> slot order: every empty slot takes the representative's rows (what running it would have given)
            yr = self.empty(x.shape[0], self.cfg.d_model)
            kn.gemm(x, a.w("img_encoder.linear.weight"), yr, live=self._lv(x.shape[0]))
            kn.rows_gather(yr, y, ip.slot_rows)
        self._ip = None
        return y, c

    def img_bwd(self, c, dy):
        a = self.arena
        n = c.n
        ip = self._ip = getattr(c, "ip", None)
        if ip is not None:     # slot order -> run order; the representative's rows are zero (its slots are masked keys: no gradient reaches them)
            dy = kn.rows_gather(dy, self.empty(dy.shape[0], dy.shape[1]), ip.run_rows, live=self._lv(dy.shape[0]))
        self.wgrad(dy, c.feat, "img_encoder.linear.weight", live=self._lv(dy.shape[0]))
        self.touch("img_encoder.linear.weight")
        dx = self.empty(c.feat.shape[0], c.feat.shape[1])
        self.dgrad(dy, "img_encoder.linear.weight", a.w("img_encoder.linear.weight"), dx, live=self._lv(dy.shape[0]))
        padded = {}
        for bc in reversed(c.blocks):
            b = bc.name
            first_block = bc.down      # block 0: its input is the detached stage-2 output (:33) -> no input gradient
            R2 = bc.o2.shape[0]
            didt = self.empty(dx.shape[0], dx.shape[1])
            dc3 = self._bn_bwd(bc.bn3, dx, dresidual=didt)
            w3 = a.w(b + "conv3.weight")
            self.wgrad(dc3, bc.o2, gview=a.g(b + "conv3.weight", (w3.shape[0], w3.shape[1])), live=self._lv(R2))
            do2 = self.empty(R2, bc.pl)
            self.dgrad(dc3, b + "conv3.weight", w3.view(w3.shape[0], w3.shape[1]), do2, live=self._lv(R2))
            wm = self.conv_mats[b + "conv2.weight"]
            wr = self.conv_mats_r.get(b + "conv2.weight") if bc.col is None else None
            if wr is not None:
                assert not first_block
                # forward ran as an implicit GEMM and the channel counts fit: no im2col matrix in the backward either.  bn2's backward writes
                # dc2 PADDED (one zero-bordered buffer serves every block of the stage: only the interior is ever written); the weight gradient
                # is the reduction-major product of the two padded images with a per-tap row shift, the input gradient the forward's implicit
                # convolution of padded dc2 with the rotated weights
                key = (n, bc.H, bc.W, bc.pl)
                dc2p = padded.get(key)
                if dc2p is None:
                    dc2p = padded[key] = self.zeros(n * (bc.H + 2) * (bc.W + 2), bc.pl)
                self._bn_bwd(bc.bn2, do2, dx_padded=dc2p, dx_pad_hw=(bc.H, bc.W))
                dwm = self.empty(wm.shape[0], 9 * bc.pl, dtype=torch.float32)
                sk = self.splitk(bc.pl, 9 * bc.pl, dc2p.shape[0], tile256=True)
                lvp = self._lv(dc2p.shape[0], -2 * (bc.W + 3))         # the reduction's length over the images that run
                if sk > 1:
                    ws = self.empty(sk * wm.shape[0], 9 * bc.pl, dtype=torch.float32)
                    kn.conv3x3_wgrad(dc2p, bc.o1, ws, n, bc.H, bc.W, bc.pl, sk, live=lvp)
                    kn.slab_reduce(ws, sk, dwm, accumulate=False)
                else:
                    kn.conv3x3_wgrad(dc2p, bc.o1, dwm, n, bc.H, bc.W, bc.pl, 1, live=lvp)
                kn.conv_matrix_grad_to_weight(dwm, a.g(b + "conv2.weight"), bc.pl, bc.pl, 3, 3, 9 * bc.pl, True)
                do1 = self.empty(n * bc.H * bc.W, bc.pl)
                kn.conv3x3_gemm(dc2p, wr, do1, n, bc.H, bc.W, bc.pl, live=self._lv(do1.shape[0]))
                dc1 = self._bn_bwd(bc.bn1, do1)
                w1 = a.w(b + "conv1.weight")
                self.wgrad(dc1, bc.x, gview=a.g(b + "conv1.weight", (w1.shape[0], w1.shape[1])), live=self._lv(dc1.shape[0]))
                self.touch(b + "conv1.weight", b + "conv2.weight", b + "conv3.weight")
                self.dgrad(dc1, b + "conv1.weight", w1.view(w1.shape[0], w1.shape[1]), didt, accumulate=True,
                           live=self._lv(dc1.shape[0]))      # never the stage's first block (stride 2)
                dx = didt
                continue
            dc2 = self._bn_bwd(bc.bn2, do2)
            dwm = self.zeros(wm.shape[0], wm.shape[1], dtype=torch.float32)
            col = bc.col
            if col is None:        # the forward ran as an implicit GEMM: the im2col matrix of the PADDED o1 (an (H+2) x (W+2) image, padding 0)
                col = self.empty(R2, wm.shape[1])
                kn.im2col(bc.o1, col, n, bc.H + 2, bc.W + 2, bc.pl, 3, 3, 1, 0, bc.H, bc.W, wm.shape[1], images=ip)
            self.wgrad(dc2, col, gview=dwm, live=self._lv(R2))
            kn.conv_matrix_grad_to_weight(dwm, a.g(b + "conv2.weight"), bc.pl, bc.pl, 3, 3, wm.shape[1], True)
            dcol = self.empty(R2, wm.shape[1])
            wmt = self.conv_mats_t.get(b + "conv2.weight")
            if wmt is not None:
                kn.gemm(dc2, wmt, dcol, live=self._lv(R2))
            else:
                kn.gemm(dc2, wm, dcol, b_t=True, live=self._lv(R2))
            do1 = self.empty(n * bc.H * bc.W, bc.pl)
            Ho2, Wo2 = (bc.H + 2 - 3) // bc.stride + 1, (bc.W + 2 - 3) // bc.stride + 1
            kn.col2im(dcol, do1, n, bc.H, bc.W, bc.pl, 3, 3, bc.stride, 1, Ho2, Wo2, wm.shape[1], images=ip)
            dc1 = self._bn_bwd(bc.bn1, do1)
            w1 = a.w(b + "conv1.weight")
            self.wgrad(dc1, bc.x, gview=a.g(b + "conv1.weight", (w1.shape[0], w1.shape[1])), live=self._lv(dc1.shape[0]))
            self.touch(b + "conv1.weight", b + "conv2.weight", b + "conv3.weight")
            if first_block:
                dcd = self._bn_bwd(bc.bnd, didt)
                wd = a.w(b + "downsample.0.weight")
                self.wgrad(dcd, bc.xs, gview=a.g(b + "downsample.0.weight", (wd.shape[0], wd.shape[1])), live=self._lv(dcd.shape[0]))
                self.touch(b + "downsample.0.weight")
                dx = None
            else:
                self.dgrad(dc1, b + "conv1.weight", w1.view(w1.shape[0], w1.shape[1]), didt, accumulate=True, live=self._lv(dc1.shape[0]))
                dx = didt
        self._ip = None
