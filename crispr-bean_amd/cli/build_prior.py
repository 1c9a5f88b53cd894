"""``bean build-prior``: prior parameters for the second of two batched runs whose libraries share
variants but no guides (``bean/cli/build_prior.py:10-66``, arguments as
``bean/model/parser_prior.py``).  The posterior of the shared variants in the first run
(``--save-raw`` pickle) becomes their prior in the second; variants new to the second run keep the
default prior."""
from __future__ import annotations

import pickle as pkl

import numpy as np
import torch

from ..model.parser import parse_args as run_parser
from ..model.run import _get_guide_target_info


def attach_args(parser):
    parser.add_argument("command1", type=str, help="bean run command for the first batched run.")
    parser.add_argument("command2", type=str, help="bean run command for the second batched run.")
    parser.add_argument("raw_run_output1", type=str,
                        help="bean run output .pkl path for the first batched run, which should be ran with --save-raw")
    parser.add_argument("output_path", type=str, help="Output path to save prior parameters.")
    return parser


def generate_prior_data_for_disjoint_library_pair(command1: str, command2: str, output1_path: str,
                                                  prior_params_path: str):
    from .run import main as get_screendata

    with open(output1_path, "rb") as f:
        data = pkl.load(f)
    ndata = data["data"]
    parser = run_parser()
    args = parser.parse_args(command1.split("bean run ")[-1].split(" "))
    args2 = parser.parse_args(command2.split("bean run ")[-1].split(" "))
    ndata2 = get_screendata(args2, return_data=True)
    target_df = _get_guide_target_info(ndata.screen, args, cols_include=[args.negctrl_col])
    target_df2 = _get_guide_target_info(ndata2.screen, args2, cols_include=[args2.negctrl_col])
    pos2 = {name: i for i, name in enumerate(target_df2.index)}
    batch1_idx = np.array([i for i, name in enumerate(target_df.index) if name in pos2], dtype=np.int64)
    batch2_idx = np.array([pos2[target_df.index[i]] for i in batch1_idx], dtype=np.int64)
    T2 = ndata2.n_targets
    params = {k: torch.as_tensor(v).reshape(-1, 1).float() for k, v in data["params"].items()
              if k in ("mu_loc", "mu_scale", "sd_loc", "sd_scale")}

    def carry(name, default):
        out = torch.full((T2, 1), default)
        out[batch2_idx, :] = params[name][batch1_idx, :]
        return out

    prior_params = {"mu_loc": carry("mu_loc", 0.0), "mu_scale": carry("mu_scale", 1.0)}
    if getattr(ndata, "selection", "sorting") == "sorting":
        prior_params["sd_loc"] = carry("sd_loc", 0.0)
        prior_params["sd_scale"] = carry("sd_scale", 0.01)
    with open(prior_params_path, "wb") as f:
        pkl.dump(prior_params, f)
    print(f"Successfully generated prior parameters at {prior_params_path}. To use this parameter, run:\n"
          f"bean run {command2.split('bean run ')[-1] + ' --prior-params ' + prior_params_path}")
    return prior_params


def main(args):
    generate_prior_data_for_disjoint_library_pair(args.command1, args.command2, args.raw_run_output1,
                                                  args.output_path)
