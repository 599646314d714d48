"""psvo_amd -- MI355X-native implementation of PSVO's SMC forward-filtering / backward-simulation
path behind the reference's own Python interface (runner_flag.py -> runner.main -> trainer ->
<Objective>.get_log_ZSMC).  Compute lives in psvo_amd/csrc (hand-written gfx950 HIP kernels
behind the C ABI of include/psvo_hip.h); this package is the host-side mirror."""
__version__ = "0.1.0"
