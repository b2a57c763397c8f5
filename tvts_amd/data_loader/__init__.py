from .synthetic import SyntheticTextVideoLoader, synth_batch, synth_batch_v1  # noqa: F401
