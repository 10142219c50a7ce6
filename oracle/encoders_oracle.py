"""TEST INFRASTRUCTURE -- CPU oracle for the two modality encoders of the hot path.

* `yelp_table_encoder`  restates /root/reference/src/table_encoder.py:5-83 (pinned against the
  imported reference by tests/golden/table_*.npz).
* `resnet101_features`  restates /root/reference/src/img_encoder.py:5-41.  The arithmetic of the
  backbone lives in a third-party dependency that is NOT under /root/reference:
  `torchvision==0.6.1` (requirements.txt:19), `torchvision.models.resnet101`.  Its published
  definition (ResNet v1.5: Bottleneck with stride on the 3x3 conv, layers [3,4,23,3], expansion 4,
  BatchNorm eps 1e-5 momentum 0.1) is restated here from torch.nn.functional primitives.
  PARITY UNPINNED for the backbone: torchvision is not installed in the development container and
  the reference holds no test or golden vector for it; only the wrapper logic (stage split,
  `.detach()` after stage 2, flatten/transpose, bias-free Linear) is checked against the reference
  (with a stand-in backbone) in oracle/make_golden.py.

Only tests/, smoke() and bench.py's cpu_baseline may import this module.
"""
import torch
import torch.nn.functional as F

RESNET101_LAYERS = (3, 4, 23, 3)


# --------------------------------------------------------------------------------------------
# Table encoder
# --------------------------------------------------------------------------------------------
def table_param_shapes(prefix="table_encoder."):
    return {
        prefix + "rating_embedding.weight": (1024, 4),
        prefix + "hours_embedding.weight": (1024, 4),
        prefix + "fc.weight": (1024, 2048),
        prefix + "fc.bias": (1024,),
        prefix + "linear.weight": (1024, 1024),
    }


def yelp_table_encoder(sd, emb_weight, field, field_value, prefix="table_encoder."):
    """field [47,6] int64; field_value = [name [B,24], category [B,6,12], str_categorical [B,5,3],
    str_boolean [B,32,1], rating [B,4], hours [B,7,4]].  -> ([B,47,1024], [B,47] bool)."""
    name, category, str_cat, str_bool, rating, hours = field_value
    E = emb_weight.detach()  # every embedding read is under no_grad (table_encoder.py:28,35,41,50,56)

    def msum(ids, dim):
        return (F.embedding(ids, E) * ids.ne(1).unsqueeze(-1).to(E.dtype)).sum(dim=dim)

    field_name = msum(field, 1)                                           # [47,D]
    name_e = msum(name, 1).unsqueeze(1)                                   # [B,1,D]
    cat_tok = msum(category, 2)                                           # [B,6,D]
    cat_valid = category.ne(1).any(dim=-1).unsqueeze(-1).to(E.dtype)          # [B,6,1]
    cat_e = (cat_tok * cat_valid).sum(dim=1, keepdim=True) / (cat_valid.sum(dim=1, keepdim=True) + 1e-6)
    strcat_e = msum(str_cat, 2)                                           # [B,5,D]
    sb = str_bool.squeeze(-1)
    strbool_e = F.embedding(sb, E) * str_bool.ne(1).to(E.dtype)               # [B,32,D]
    rating_e = F.linear(rating.to(E.dtype), sd[prefix + "rating_embedding.weight"]).unsqueeze(1)
    hours_e = F.linear(hours.to(E.dtype), sd[prefix + "hours_embedding.weight"])
    B = name.shape[0]
    names = field_name.unsqueeze(0).expand(B, -1, -1)
    values = torch.cat([name_e, cat_e, strcat_e, strbool_e, rating_e, hours_e], dim=1)   # [B,47,D]
    x = torch.cat([names, values], dim=-1)                                # [B,47,2D]
    # through bart_oracle.linear: identical to F.linear unless the bf16 emulation is on (tests only), which then rounds these two
    # Linears like every other one of the path
    from oracle import bart_oracle as _bo
    x = _bo.linear(x, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    x = _bo.linear(torch.relu(x), sd[prefix + "linear.weight"])
    ones = torch.ones(B, 1, dtype=torch.bool)
    mask = torch.cat([ones, category[:, :1, 0].ne(1), str_cat[:, :, 0].ne(1), str_bool[:, :, 0].ne(1),
                      ones, hours.sum(dim=-1) != 0], dim=1)
    return x, mask


def amazon_table_param_shapes(prefix="table_encoder."):
    return {
        prefix + "price_embedding.weight": (1024, 11),
        prefix + "rating_embedding.weight": (1024, 4),
        prefix + "fc.weight": (1024, 2048),
        prefix + "fc.bias": (1024,),
        prefix + "linear.weight": (1024, 1024),
    }


def amazon_table_encoder(sd, emb_weight, field, field_value, prefix="table_encoder."):
    """AmazonTableEncoder.forward (/root/reference/src/table_encoder.py:94-167).  field [6,1] int64; field_value =
    [price [B,11], rating [B,4], brand [B,12], name [B,32], category [B,3,8,12], description [B,128]].
    -> ([B,133,1024], [B,133] bool)."""
    price, rating, brand, name, category, description = field_value
    E = emb_weight.detach()                                               # embedding reads are under no_grad (:108,119,125,131,147)

    def msum(ids, dim):
        return (F.embedding(ids, E) * ids.ne(1).unsqueeze(-1).to(E.dtype)).sum(dim=dim)

    fn = F.embedding(field, E).squeeze(1)                                 # [6,D]
    field_name = torch.cat([fn[:-1], fn[-1:].repeat(128, 1)])             # [133,D]  (:110)
    price_e = F.linear(price.to(E.dtype), sd[prefix + "price_embedding.weight"]).unsqueeze(1)
    rating_e = F.linear(rating.to(E.dtype), sd[prefix + "rating_embedding.weight"]).unsqueeze(1)
    brand_e = msum(brand, 1).unsqueeze(1)
    name_e = msum(name, 1).unsqueeze(1)
    rows = msum(category, 3)                                              # [B,3,8,D]
    row_valid = category.ne(1).any(dim=-1)                                # [B,3,8]
    rv = row_valid.unsqueeze(-1).to(E.dtype)
    groups = (rows * rv).sum(dim=2) / (rv.sum(dim=2) + 1e-6)              # [B,3,D]
    gv = row_valid.any(dim=-1).unsqueeze(-1).to(E.dtype)                      # [B,3,1]
    cat_e = (groups * gv).sum(dim=1, keepdim=True) / (gv.sum(dim=1, keepdim=True) + 1e-6)
    desc_e = F.embedding(description, E)                                  # not masked (:148)
    B = price.shape[0]
    names = field_name.unsqueeze(0).expand(B, -1, -1)
    values = torch.cat([price_e, rating_e, brand_e, name_e, cat_e, desc_e], dim=1)       # [B,133,D]
    x = torch.cat([names, values], dim=-1)
    # through bart_oracle.linear: identical to F.linear unless the bf16 emulation is on (tests only), which then rounds these two
    # Linears like every other one of the path
    from oracle import bart_oracle as _bo
    x = _bo.linear(x, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    x = _bo.linear(torch.relu(x), sd[prefix + "linear.weight"])
    ones = torch.ones(B, 1, dtype=torch.bool)
    mask = torch.cat([price.sum(dim=1, keepdim=True) != 0, ones, brand[:, :1].ne(1), name[:, :1].ne(1), ones, description.ne(1)], dim=1)
    return x, mask


# --------------------------------------------------------------------------------------------
# ResNet101 (torchvision 0.6.1 definition), stages 1-3 + projection
# --------------------------------------------------------------------------------------------
def resnet_param_shapes(embedding_dim=1024, prefix="img_encoder."):
    """Keys as `Resnet.state_dict()` dumps them under `resnet.*` (the aliased `stage{1,2,3}.*`
    entries point at the same tensors and are not repeated here).  layer4/fc are registered by the
    reference (img_encoder.py:8-24) but never called; they are listed so checkpoints round-trip."""
    r = prefix + "resnet."
    s = {}

    def bn(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)
        s[name + ".running_mean"] = (c,)
        s[name + ".running_var"] = (c,)
        s[name + ".num_batches_tracked"] = ()

    s[r + "conv1.weight"] = (64, 3, 7, 7)
    bn(r + "bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), RESNET101_LAYERS)):
        for bi in range(blocks):
            b = r + "layer%d.%d." % (li + 1, bi)
            s[b + "conv1.weight"] = (planes, inplanes, 1, 1)
            bn(b + "bn1", planes)
            s[b + "conv2.weight"] = (planes, planes, 3, 3)
            bn(b + "bn2", planes)
            s[b + "conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(b + "bn3", planes * 4)
            if bi == 0:
                s[b + "downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                bn(b + "downsample.1", planes * 4)
            inplanes = planes * 4
    s[r + "fc.weight"] = (1000, 2048)
    s[r + "fc.bias"] = (1000,)
    s[prefix + "linear.weight"] = (embedding_dim, 1024)
    return s


def _q(t):
    """bf16 emulation (tests only, see bart_oracle.EMULATE_BF16): round a tensor to bf16 in the forward pass, identity gradient.
    Off = the reference's arithmetic untouched."""
    from oracle import bart_oracle as _bo
    if not _bo.EMULATE_BF16:
        return t
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


def _conv(x, w, **kw):
    return _q(F.conv2d(_q(x), _q(w), **kw))


def _bn(sd, name, x, training, running):
    return _q(_bn_exact(sd, name, x, training, running))


def _bn_exact(sd, name, x, training, running):
    """BatchNorm2d, eps 1e-5, momentum 0.1.  `running` (dict) receives updated running stats in
    train mode, mirroring nn.BatchNorm2d's buffer side effect."""
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
    if training:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, 0.1, 1e-5)
        if running is not None:
            running[name + ".running_mean"] = rm2
            running[name + ".running_var"] = rv2
        return y
    return F.batch_norm(x, rm, rv, w, b, False, 0.1, 1e-5)


def _bottleneck(sd, b, x, stride, has_down, training, running):
    idt = x
    o = F.relu(_bn(sd, b + "bn1", _conv(x, sd[b + "conv1.weight"]), training, running))
    o = F.relu(_bn(sd, b + "bn2", _conv(o, sd[b + "conv2.weight"], stride=stride, padding=1), training, running))
    o = _bn(sd, b + "bn3", _conv(o, sd[b + "conv3.weight"]), training, running)
    if has_down:
        idt = _bn(sd, b + "downsample.1", _conv(x, sd[b + "downsample.0.weight"], stride=stride), training, running)
    return F.relu(o + idt)


def _layer(sd, r, li, x, training, running):
    blocks = RESNET101_LAYERS[li - 1]
    for bi in range(blocks):
        stride = 2 if (bi == 0 and li > 1) else 1
        x = _bottleneck(sd, r + "layer%d.%d." % (li, bi), x, stride, bi == 0, training, running)
    return x


def resnet101_features(sd, x, training=True, running=None, prefix="img_encoder."):
    """Resnet.forward (img_encoder.py:31-41): x [n,3,224,224] -> [n,196,D].  Stage-2 output is
    detached (:33) so stem/layer1/layer2 get no gradient; BN runs in batch-statistics mode when
    `training` (zero-padded images included -- SURVEY.md section 7 hard parts)."""
    r = prefix + "resnet."
    x = _conv(x, sd[r + "conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(sd, r + "bn1", x, training, running))
    x = F.max_pool2d(x, 3, 2, 1)
    x = _layer(sd, r, 1, x, training, running)
    x = _layer(sd, r, 2, x, training, running).detach()
    x = _layer(sd, r, 3, x, training, running)
    x = x.flatten(start_dim=-2).transpose(1, 2)             # [n, HW, 1024]
    from oracle import bart_oracle as _bo
    return _bo.linear(x, sd[prefix + "linear.weight"])
