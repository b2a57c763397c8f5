"""Drop-in for v2/downstream/model_TVTSv2_ViT_H_14.py: same class name, constructor and forward contract."""
from ._common import DownstreamBase, sim_matrix  # noqa: F401


class TVTSv2_H_14(DownstreamBase):
    ARCH_NAME = "H_14"
