"""Where a run's results go and how its parameters are recorded.

Same on-disk contract as the reference (src/rslts_saving/rslts_saving.py:14-60, src/rslts_saving/datetools.py:11-21), so that
its notebooks find what they look for (notebooks/PSVO.ipynb picks the newest directory under rslts/<rslt_dir_name>/ and reads
history.json and param.json from it):

    <cwd>/rslts/<rslt_dir_name>/D<yymmdd>_<HHMMSS>_<key>_<value>_<key>_<value>.../

with the key/value pairs in the order of the dict the runner passes (np, t, bs, lr, epoch, seed) and `rslt_dir_name` itself
left out of the suffix; `param.json` maps every flag name to str(value); numpy scalars / arrays inside history.json are
written as plain JSON numbers / lists.  The plotting half of the reference module (matplotlib, seaborn) is presentation and
is not built.
"""
import datetime
import json
import os

import numpy as np


def run_stamp(now=None):
    """'D<yymmdd>_<HHMMSS>' -- the reference's addDateTime() without its leading underscore"""
    now = now or datetime.datetime.now()
    return now.strftime("D%y%m%d_%H%M%S")


def addDateTime(s=""):
    """reference spelling (datetools.py): the stamp appended to `s` behind an underscore"""
    return s + "_" + run_stamp()


def create_RLT_DIR(Experiment_params):
    """Make (and return, with a trailing slash) the result directory of one run; see the module docstring for the name."""
    suffix = "".join("_{}_{}".format(key, value) for key, value in Experiment_params.items() if key != "rslt_dir_name")
    cwd = os.getcwd().replace("\\", "/")
    path = "/".join([cwd, "rslts", Experiment_params["rslt_dir_name"], run_stamp() + suffix]) + "/"
    os.makedirs(path, exist_ok=True)
    return path


def save_experiment_param(RLT_DIR, FLAGS):
    """param.json: {flag name: str(value)} for every flag, names sorted; also echoed like the reference does"""
    flags = FLAGS.as_dict() if hasattr(FLAGS, "as_dict") else {k: v for k, v in vars(FLAGS).items() if not k.startswith("_")}
    record = {name: str(flags[name]) for name in sorted(flags)}
    print("Experiment_params:")
    for name, value in record.items():
        print("\t{}: {}".format(name, value))
    with open(os.path.join(RLT_DIR, "param.json"), "w") as fh:
        json.dump(record, fh, indent=4, cls=NumpyEncoder)


class NumpyEncoder(json.JSONEncoder):
    """json encoder that accepts numpy scalars and arrays (history.json holds lists of np.float64 / R-square arrays)"""

    def default(self, obj):
        if isinstance(obj, np.generic):
            return obj.item()
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        return super().default(obj)
