"""Ship the reference's default Fitzhugh-Nagumo observations and the training log its notebook holds as one committed
fixture (DATA only: observation arrays and the numbers the notebook printed).

    python tests/golden/make_fhn_slice.py          (build container: reads /root/reference, numpy pickle + ipynb JSON)

Sources
  * /root/reference/data/fhn/[1,0]_obs_cov_0.01/datadict -- the default `datadir` of src/runner_flag.py:36 and of
    notebooks/PSVO.ipynb: Ytrain (200, 200, 1), Yvalid (40, 200, 1) (float64 in the pickle, float32 here: the path
    computes in fp32).
  * /root/reference/notebooks/PSVO.ipynb, output of the `tf.app.run()` cell: PSVO, n_particles 16,
    n_particles_for_BSim_proposal 8, batch_size 1, lr 3e-3, print_freq 10, H = 32, Dh = 32, seed 0, TF 1.12.  Every
    evaluation block "iter K / Train log_ZSMC: a, valid log_ZSMC: b / Train, Valid k-step Rsq: [31] [31]" is parsed
    into rows (40 evaluations: iter 1, 10, 20, ... 390; best valid cost on iter 190).  These are the only outcomes the
    reference published for the path.

Written: tests/golden/fhn_notebook.npz
    Ytrain, Yvalid                       the observations
    nb_iter (40,)                        evaluation iteration (epoch count)
    nb_train_log_ZSMC, nb_valid_log_ZSMC (40,)
    nb_train_Rsq, nb_valid_Rsq (40, 31)  k-step R-square, k = 0..30
    nb_best_valid_iter                   190
    nb_epoch_seconds (n,)                the "epoch K took S seconds" lines (TF-CPU, one i7-class desktop; context only)
Consumers: tests/test_reference_anchor.py (first line, as a sanity band), tests/notebook_curve.py and
tests/test_gpu_notebook_curve.py (the training curve)."""
import json
import os
import pickle
import re

import numpy as np

DATA = "/root/reference/data/fhn/[1,0]_obs_cov_0.01/datadict"
NOTEBOOK = "/root/reference/notebooks/PSVO.ipynb"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fhn_notebook.npz")

_NUM = r"[-+]?\d+\.\d+(?:e[-+]?\d+)?"
_BLOCK = re.compile(r"iter (\d+)\nTrain log_ZSMC: (" + _NUM + r"), valid log_ZSMC: (" + _NUM + r")\n"
                    r"Train, Valid k-step Rsq:\n \[([^\]]*)\] \n \[([^\]]*)\]")


def notebook_log(path=NOTEBOOK):
    nb = json.load(open(path))
    text = ""
    for cell in nb["cells"]:
        if cell["cell_type"] == "code" and "tf.app.run()" in "".join(cell["source"]):
            text = "".join("".join(o["text"]) for o in cell["outputs"] if "text" in o)
    rows = _BLOCK.findall(text)
    it = np.array([int(r[0]) for r in rows])
    tr = np.array([float(r[1]) for r in rows])
    va = np.array([float(r[2]) for r in rows])
    rt = np.array([[float(x) for x in r[3].split()] for r in rows])
    rv = np.array([[float(x) for x in r[4].split()] for r in rows])
    best = int(re.findall(r"best valid cost on iter: (\d+)", text)[-1])
    secs = np.array([float(s) for s in re.findall(r"epoch \d+\s+took (" + _NUM + r") seconds", text)])
    return it, tr, va, rt, rv, best, secs


if __name__ == "__main__":
    with open(DATA, "rb") as f:
        d = pickle.load(f)
    it, tr, va, rt, rv, best, secs = notebook_log()
    assert it[0] == 1 and tr[0] == -778.343 and va[0] == -775.139 and rt.shape == (len(it), 31) == rv.shape
    np.savez_compressed(DST, Ytrain=d["Ytrain"].astype(np.float32), Yvalid=d["Yvalid"].astype(np.float32),
                        nb_iter=it, nb_train_log_ZSMC=tr, nb_valid_log_ZSMC=va, nb_train_Rsq=rt, nb_valid_Rsq=rv,
                        nb_best_valid_iter=np.int64(best), nb_epoch_seconds=secs)
    print(DST, os.path.getsize(DST), "evaluations:", len(it), "epochs timed:", len(secs), "median s/epoch:", np.median(secs))
