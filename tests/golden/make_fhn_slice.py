"""Slice the reference's default Fitzhugh-Nagumo data set into a small committed fixture (DATA only: observations).

    python tests/golden/make_fhn_slice.py          (build container: reads /root/reference/data, numpy pickle)

Source: /root/reference/data/fhn/[1,0]_obs_cov_0.01/datadict (the default `datadir` of src/runner_flag.py:36 and of
notebooks/PSVO.ipynb): Ytrain (200, 200, 1), Yvalid (40, 200, 1).  Written: tests/golden/fhn_obs_slice.npz with the first
40 training and the 40 validation observation sequences (float32; 2 x 32 KB).  The notebook's first evaluation line
(cell 31: "Train log_ZSMC: -778.343, valid log_ZSMC: -775.139", fresh TF-seed-0 initialisation, N = 16, M = 8, H = 32,
Dh = 32, batch 1, T = 200) is the only number the reference holds for this path; tests/test_reference_anchor.py checks a
fresh-init model against it as a SANITY BAND (different initial weights and draws: not a parity pin)."""
import os
import pickle

import numpy as np

SRC = "/root/reference/data/fhn/[1,0]_obs_cov_0.01/datadict"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fhn_obs_slice.npz")

if __name__ == "__main__":
    with open(SRC, "rb") as f:
        d = pickle.load(f)
    np.savez_compressed(DST, Ytrain=d["Ytrain"][:40].astype(np.float32), Yvalid=d["Yvalid"][:40].astype(np.float32),
                        notebook_train_log_ZSMC=np.float64(-778.343), notebook_valid_log_ZSMC=np.float64(-775.139))
    print(DST, os.path.getsize(DST))
