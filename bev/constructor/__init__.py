from bev_amd.overlay import extend as _extend

__path__ = _extend(__path__, __name__)
__all__ = ["homo_constr"]
