"""`bev.tracker`: the geometry front-end of the reference's SORT tracker served by the MI355X path
(/root/reference/bev/tracker/rbox_tracker.py:87-92, :383-405).  The Kalman filters and the Hungarian assignment of that
file are per-track host logic and stay the reference's: `rbox_tracker` loads them from a co-installed
reference when there is one."""
from bev_amd.overlay import extend as _extend

__path__ = _extend(__path__, __name__)
__all__ = ["rbox_tracker"]
