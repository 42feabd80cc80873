"""`bev.tracker`: the geometry front-end of the reference's SORT tracker served by the MI355X path
(/root/reference/bev/tracker/rbox_tracker.py:87-92, :383-405).  The Kalman filters and the Hungarian assignment of that
file are per-track host logic and stay out of scope."""
