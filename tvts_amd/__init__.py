"""tvts_amd -- MI355X-native TVTSv2 pretrain step (hand-written gfx950 HIP kernels behind the
reference's model / loss / trainer surface).  See DESIGN.md and INTEGRATION.md."""
__all__ = ["arch", "engine", "hip"]
