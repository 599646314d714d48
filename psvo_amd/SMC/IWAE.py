"""IWAE -- mirror of reference src/SMC/IWAE.py:8-12."""
from .SVO import SVO


class IWAE(SVO):
    def __init__(self, model, FLAGS, name="log_ZSMC"):
        SVO.__init__(self, model, FLAGS, name)
        self.resample_particles = False
        self.smooth_obs = False
