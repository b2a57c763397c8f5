"""Drop-in for v2/downstream/model_TVTSv2_ViT_B_16.py: same class name, constructor and forward contract."""
from ._common import DownstreamBase, sim_matrix  # noqa: F401


class TVTSv2_B_16(DownstreamBase):
    ARCH_NAME = "B_16"
