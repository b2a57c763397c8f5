from .synthetic import SyntheticTextVideoLoader, synth_batch  # noqa: F401
