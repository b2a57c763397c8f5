from .synthetic import SyntheticTextVideoLoader, synth_batch, synth_batch_v1  # noqa: F401
from .transforms import CaptionCache, pil_nearest_table, resize_sizes, resize_tables  # noqa: F401
