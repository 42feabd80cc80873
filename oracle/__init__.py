"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/warp_oracle.c header).

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package `bev_amd` never imports this.
"""
