"""AESMC -- mirror of reference src/SMC/AESMC.py:8-11."""
from .SVO import SVO


class AESMC(SVO):
    def __init__(self, model, FLAGS, name="log_ZSMC"):
        SVO.__init__(self, model, FLAGS, name)
        self.smooth_obs = False
