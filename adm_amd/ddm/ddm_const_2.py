"""HIP drop-in for /root/reference/ddm/ddm_const_2.py: x_t = x0 + C t + t eps."""
from .ddpm import DDPMBase


class DDPM(DDPMBase):
    SCHEDULE = "const_2"
    DEFAULT_EPS = 1e-4
