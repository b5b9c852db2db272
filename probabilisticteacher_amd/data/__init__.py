from .augment import (StrongParams, hflip_batch, resize_batch, resize_shortest_edge_size, sample_strong_params,  # noqa: F401
                      strong_augment_batch)
from .mapper import AspectRatioGroupedSemiSupDatasetTwoCrop, DeviceTwoCropMapper  # noqa: F401
