"""multimodalsum_amd -- MI355X-native (gfx950) implementation of the MultimodalSum training hot path.

Importing the package loads libmmsum_hip.so (the hand-written HIP kernels behind a C ABI, see
include/mmsum_hip.h); there is no CPU or PyTorch-op fallback for the compute path.
"""
from . import _lib  # noqa: F401  (raises if the HIP library is missing)
from .config import BartConfig  # noqa: F401
from .modules import (AmazonTableEncoder, BartForEncConditionalGeneration, BartForMultiEncConditionalGeneration, ImgSupervised,  # noqa: F401
                      LabelSmoothingLoss, MultimodalSum, Resnet, TableSupervised, TextSupervised, YelpTableEncoder)
from .optim import FusedAdamW, clip_grad_norm_, get_linear_schedule_with_warmup, get_optimizer  # noqa: F401
from .parallel import DistributedDataParallel, reduce_tensor  # noqa: F401
from .prefetch import amazon_data_prefetcher, data_prefetcher, yelp_data_prefetcher  # noqa: F401
