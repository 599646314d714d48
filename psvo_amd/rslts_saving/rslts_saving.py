"""Result-directory helpers: mirror of the non-plotting part of reference
src/rslts_saving/rslts_saving.py:14-47 (directory naming, param.json, NumpyEncoder).  The
matplotlib / seaborn plots are presentation only and out of the MI355X hot-path scope."""
import json
import os
import time

import numpy as np


def addDateTime(s=""):
    """reference src/rslts_saving/datetools.py: timestamp suffix"""
    return s + time.strftime("%y%m%d%H%M%S")


def create_RLT_DIR(Experiment_params):
    # create the dir to save data
    cur_date = addDateTime()
    local_rlt_root = "rslts/" + Experiment_params["rslt_dir_name"] + "/"
    params_str = ""
    for param_name, param in Experiment_params.items():
        if param_name == "rslt_dir_name":
            continue
        params_str += "_" + param_name + "_" + str(param)
    RLT_DIR = os.getcwd().replace("\\", "/") + "/" + local_rlt_root + cur_date + params_str + "/"
    if not os.path.exists(RLT_DIR):
        os.makedirs(RLT_DIR)
    return RLT_DIR


def save_experiment_param(RLT_DIR, FLAGS):
    params_dict = {}
    params_list = sorted([param for param in dir(FLAGS) if not param.startswith("_") and param != "as_dict"])
    for param in params_list:
        params_dict[param] = str(getattr(FLAGS, param))
    with open(RLT_DIR + "param.json", "w") as f:
        json.dump(params_dict, f, indent=4, cls=NumpyEncoder)


class NumpyEncoder(json.JSONEncoder):
    def default(self, obj):
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        if isinstance(obj, (np.floating, np.integer)):
            return obj.item()
        return json.JSONEncoder.default(self, obj)
