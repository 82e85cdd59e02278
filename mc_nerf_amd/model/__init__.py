"""Same export surface as the reference's ``model/__init__.py`` for the hot path."""
from .mc_nerf import MC_Model, NeRF_Model
from .net_block import CorseFine_NeRF, SinCosEmbedding
from .net_utils import RAdam
from .loss import MC_NeRF_Loss

__all__ = ["MC_Model", "NeRF_Model", "CorseFine_NeRF", "SinCosEmbedding", "RAdam", "MC_NeRF_Loss"]
