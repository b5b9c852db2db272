from .augment import (StrongParams, hflip_batch, resize_batch, resize_shortest_edge_size, sample_strong_params,  # noqa: F401
                      strong_augment_batch)
from .mapper import AspectRatioGroupedSemiSupDatasetTwoCrop, DeviceTwoCropMapper  # noqa: F401
from .build import build_detection_semisup_train_loader_two_crops, build_detection_test_loader, training_sampler  # noqa: F401,E402
from . import datasets  # noqa: F401,E402
