"""Synthetic Yelp-shaped batches.

The reference's data layer (/root/reference/src/data_utils.py:48-88,
/root/reference/src/multimodal_train.py:85-108) is host-side ETL and out of scope; this module
produces tensors with exactly its output contract (shapes, dtypes, pad conventions) from a seed,
on the CPU generator so the CPU oracle and the GPU path see identical inputs.
"""
import torch

PAD, BOS, EOS = 1, 0, 2


def _gen(seed):
    g = torch.Generator()
    g.manual_seed(int(seed))
    return g


def token_batch(rows, seq_len, vocab, seed, min_len=None, mean_len=None, std_len=None, g=None):
    """[rows, seq_len] int64: random ids in [3, vocab), EOS at position len-1 when len < seq_len,
    PAD(1) after.  BOS is stripped by the reference (data_utils.py:50), so none appears."""
    g = g or _gen(seed)
    min_len = min_len if min_len is not None else max(2, seq_len // 4)
    mean_len = mean_len if mean_len is not None else 0.6 * seq_len
    std_len = std_len if std_len is not None else 0.16 * seq_len
    lens = torch.clamp(torch.round(torch.randn(rows, generator=g) * std_len + mean_len), min_len, seq_len).long()
    ids = torch.randint(3, vocab, (rows, seq_len), generator=g)
    pos = torch.arange(seq_len).unsqueeze(0)
    ids = torch.where(pos == (lens - 1).unsqueeze(1), torch.where(lens.unsqueeze(1) < seq_len,
                      torch.full_like(ids, EOS), ids), ids)
    ids = torch.where(pos >= lens.unsqueeze(1), torch.full_like(ids, PAD), ids)
    return ids


def _trailing_pad(shape, vocab, g, min_real=0):
    """ids with a random number (>= min_real) of real tokens followed by pads along the last dim."""
    L = shape[-1]
    ids = torch.randint(3, vocab, shape, generator=g)
    n_real = torch.randint(min_real, L + 1, shape[:-1] + (1,), generator=g)
    return torch.where(torch.arange(L).expand(shape) < n_real, ids, torch.full_like(ids, PAD))


def table_batch(B, vocab, seed, g=None):
    """field [47,6] + the six Yelp field_value tensors (data_utils.py:67-88)."""
    g = g or _gen(seed)
    field = _trailing_pad((47, 6), vocab, _gen(977), min_real=2)   # global constant (multimodal_train.py:59-60)
    name = _trailing_pad((B, 24), vocab, g, min_real=1)
    category = _trailing_pad((B, 6, 12), vocab, g, min_real=1)
    drop_rows = torch.rand(B, 6, generator=g) < 0.5
    drop_rows[:, :3] = False
    category = torch.where(drop_rows.unsqueeze(-1), torch.full_like(category, PAD), category)
    if B > 1:
        category[B - 1] = PAD                                       # a business with no category at all
    str_cat = _trailing_pad((B, 5, 3), vocab, g, min_real=0)
    str_bool = torch.randint(3, vocab, (B, 32, 1), generator=g)
    str_bool = torch.where(torch.rand(B, 32, 1, generator=g) < 0.3, torch.full_like(str_bool, PAD), str_bool)
    rating = (torch.rand(B, 4, generator=g) < 0.5).long()
    day = torch.randint(0, 4, (B, 7), generator=g)
    hours = torch.nn.functional.one_hot(day, 4).long()
    hours = torch.where((torch.rand(B, 7, generator=g) < 0.2).unsqueeze(-1), torch.zeros_like(hours), hours)
    return field, [name, category, str_cat, str_bool, rating, hours]


def amazon_table_batch(B, vocab, seed, g=None):
    """field [6,1] + the six Amazon field_value tensors (data_utils.py:90-116): price one-hot over 11 bins (all-zero =
    unknown), rating 4 bits, brand [B,12], name [B,32], category [B,3,8,12] (groups / rows trailing-padded), description
    [B,128]; the last business has no brand, no category and no description."""
    g = g or _gen(seed)
    field = torch.randint(3, vocab, (6, 1), generator=_gen(978))
    price = torch.nn.functional.one_hot(torch.randint(0, 11, (B,), generator=g), 11).long()
    price = torch.where((torch.rand(B, generator=g) < 0.25).unsqueeze(-1), torch.zeros_like(price), price)
    rating = (torch.rand(B, 4, generator=g) < 0.5).long()
    brand = _trailing_pad((B, 12), vocab, g, min_real=1)
    name = _trailing_pad((B, 32), vocab, g, min_real=1)
    category = _trailing_pad((B, 3, 8, 12), vocab, g, min_real=1)
    rows_real = torch.randint(1, 9, (B, 3), generator=g)
    category = torch.where((torch.arange(8).view(1, 1, 8) >= rows_real.unsqueeze(-1)).unsqueeze(-1), torch.full_like(category, PAD), category)
    groups_real = torch.randint(1, 4, (B,), generator=g)
    category = torch.where((torch.arange(3).view(1, 3) >= groups_real.unsqueeze(-1)).view(B, 3, 1, 1), torch.full_like(category, PAD), category)
    description = _trailing_pad((B, 128), vocab, g, min_real=0)
    if B > 1:
        brand[B - 1] = PAD
        category[B - 1] = PAD
        description[B - 1] = PAD
    return field, [price, rating, brand, name, category, description]


def yelp_batch(B, NR, S, I, vocab, seed, img_hw=224, mean_len=None, std_len=None, min_len=None):
    """One step's inputs, row 0 of SURVEY.md section 8a.  Real sizes: NR=9, S=128, I=4, img_hw=224,
    review length ~ clamp(round(N(75,20)),32,128)."""
    g = _gen(seed)
    if S == 128 and mean_len is None:
        mean_len, std_len, min_len = 75.0, 20.0, 32
    reviews = token_batch(B * NR, S, vocab, seed, min_len=min_len, mean_len=mean_len, std_len=std_len, g=g).view(B, NR, S)
    reviews_mask = reviews.ne(PAD).long()
    rating = torch.randint(1, 6, (B, NR), generator=g).float()
    field, fv = table_batch(B, vocab, seed, g=g)
    img = torch.randn(B, I, 3, img_hw, img_hw, generator=g)
    n_valid = torch.randint(0, I + 1, (B,), generator=g)
    img_mask = torch.arange(I).unsqueeze(0) < n_valid.unsqueeze(1)
    img = img * img_mask[:, :, None, None, None].float()           # missing slots are real zeros
    return {"reviews": reviews, "reviews_mask": reviews_mask, "reviews_rating": rating, "field": field,
            "field_value": fv, "img": img, "img_mask": img_mask}


def batch_to(batch, device):
    out = {}
    for k, v in batch.items():
        out[k] = [t.to(device) for t in v] if isinstance(v, list) else v.to(device)
    return out


class DeviceBatches:
    """Yelp-shaped training batches generated ON THE DEVICE, a new one per call (inputs are regenerated
    every step so that the loader is not measured, and every step sees fresh review lengths / image counts -- which is what
    exercises the device-side live row counts of the fused step).  Same distributions as `yelp_batch`; the values differ
    (device generator), which is irrelevant for a throughput run.  The constant `field` tensor is made once."""

    def __init__(self, B, NR, S, I, vocab, device, seed, img_hw=224, no_images=False):
        """no_images: BASELINE config 3 (text + table): every image slot is zeros and img_mask is False everywhere."""
        self.B, self.NR, self.S, self.I, self.vocab, self.img_hw = B, NR, S, I, vocab, img_hw
        self.no_images = no_images
        self.device = torch.device(device)
        self.g = torch.Generator(device=self.device)
        self.g.manual_seed(int(seed))
        self.field = table_batch(1, vocab, seed)[0].to(self.device)
        self.pos = torch.arange(S, device=self.device).unsqueeze(0)

    def _rand(self, *shape):
        return torch.rand(*shape, generator=self.g, device=self.device)

    def _ids(self, *shape):
        return torch.randint(3, self.vocab, shape, generator=self.g, device=self.device)

    def _trailing(self, shape, min_real=0):
        L = shape[-1]
        n_real = torch.randint(min_real, L + 1, shape[:-1] + (1,), generator=self.g, device=self.device)
        ids = self._ids(*shape)
        return torch.where(torch.arange(L, device=self.device).expand(shape) < n_real, ids, torch.full_like(ids, PAD))

    def next(self):
        B, NR, S, I, dev = self.B, self.NR, self.S, self.I, self.device
        mean_len, std_len, min_len = (75.0, 20.0, 32) if S == 128 else (0.6 * S, 0.16 * S, max(2, S // 4))
        lens = torch.clamp(torch.round(torch.randn(B * NR, generator=self.g, device=dev) * std_len + mean_len), min_len, S).long().unsqueeze(1)
        ids = self._ids(B * NR, S)
        ids = torch.where((self.pos == lens - 1) & (lens < S), torch.full_like(ids, EOS), ids)
        reviews = torch.where(self.pos >= lens, torch.full_like(ids, PAD), ids).view(B, NR, S)
        rating = torch.randint(1, 6, (B, NR), generator=self.g, device=dev).float()
        name = self._trailing((B, 24), 1)
        category = self._trailing((B, 6, 12), 1)
        drop = self._rand(B, 6) < 0.5
        drop[:, :3] = False
        category = torch.where(drop.unsqueeze(-1), torch.full_like(category, PAD), category)
        str_cat = self._trailing((B, 5, 3), 0)
        str_bool = self._ids(B, 32, 1)
        str_bool = torch.where(self._rand(B, 32, 1) < 0.3, torch.full_like(str_bool, PAD), str_bool)
        rbits = (self._rand(B, 4) < 0.5).long()
        day = torch.randint(0, 4, (B, 7), generator=self.g, device=dev)
        hours = torch.nn.functional.one_hot(day, 4).long()
        hours = torch.where((self._rand(B, 7) < 0.2).unsqueeze(-1), torch.zeros_like(hours), hours)
        img = torch.randn(B, I, 3, self.img_hw, self.img_hw, generator=self.g, device=dev)
        n_valid = torch.randint(0, I + 1, (B,), generator=self.g, device=dev)
        if self.no_images:
            n_valid = torch.zeros_like(n_valid)
        img_mask = torch.arange(I, device=dev).unsqueeze(0) < n_valid.unsqueeze(1)
        img = img * img_mask[:, :, None, None, None].float()
        return {"reviews": reviews, "reviews_mask": reviews.ne(PAD).long(), "reviews_rating": rating, "field": self.field,
                "field_value": [name, category, str_cat, str_bool, rbits, hours], "img": img, "img_mask": img_mask}
