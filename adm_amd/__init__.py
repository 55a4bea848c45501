"""adm_amd: MI355X-native (gfx950) hot path of DDM -- UNet forward/backward + analytic-schedule loop.

Python here is host glue over libadm_hip.so (see include/adm_hip.h); there is no CPU / PyTorch fallback.
"""
__version__ = "0.1.0"
