"""HIP drop-in for /root/reference/ddm/ddm_const.py: x_t = x0 + C t + sqrt(t) eps (eps default 1e-4)."""
from .ddpm import DDPMBase


class DDPM(DDPMBase):
    SCHEDULE = "const"
    DEFAULT_EPS = 1e-4
