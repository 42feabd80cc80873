"""`bev.tool`: `compo` is served by the MI355X path; `io_vis` and the tracker tools fall through to the reference."""
from bev_amd.overlay import extend as _extend

__path__ = _extend(__path__, __name__)
__all__ = ["compo"]
