from .augment import StrongParams, sample_strong_params, strong_augment_batch, hflip_batch  # noqa: F401
from .mapper import AspectRatioGroupedSemiSupDatasetTwoCrop, DeviceTwoCropMapper  # noqa: F401
