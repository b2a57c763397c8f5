"""Drop-in for v2/model/model_dist_TVTSv2_ViT_H_14.py: same class name, ctor and forward contract."""
from ._common import TVTSv2Base, sim_matrix  # noqa: F401


class TVTSv2_H_14(TVTSv2Base):
    ARCH_NAME = "H_14"

