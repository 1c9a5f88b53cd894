"""``bean run``: screen -> tensors -> (negative-control fit) -> main fit -> tables.

Follows ``bean/cli/run.py:66-311`` step for step (variant and tiling library designs);
every step delegates to the module that mirrors the reference's.  The fit itself
runs on the MI355X through ``run_inference`` (``libbean_hip``).
"""
from __future__ import annotations

import logging
import os
import pickle as pkl
from copy import deepcopy
from functools import partial

import numpy as np

from ..framework import read_h5ad
from ..model.readwrite import write_result_table
from ..model.run import (_check_prior_params, _get_guide_info, _get_guide_target_info, check_args,
                         identify_model_guide, identify_negctrl_model_guide, run_inference)
from ..model.tiling_info import guide_to_variant_df, variant_table
from ..preprocessing.screen_data import DATACLASS_DICT
from ..preprocessing.utils import prepare_bdata

logging.basicConfig(level=logging.INFO, format="%(levelname)-5s @ %(asctime)s:\n\t %(message)s \n",
                    datefmt="%a, %d %b %Y %H:%M:%S")
logger = logging.getLogger("bean_run")
info, warn = logger.info, logger.warning


def _init_distributed():
    """`torchrun --nproc-per-node N bin/bean run ...`: one process per GPU; every rank builds the
    same screen, `run_inference` shards the guides, rank 0 writes the tables.  Returns (rank, world).
    BEAN_DIST_BACKEND=gloo and BEAN_DIST_SINGLE_DEVICE=1 exist for rehearsals on a one-GPU box."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    if not dist.is_initialized():
        local = 0 if os.environ.get("BEAN_DIST_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        backend = os.environ.get("BEAN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def main(args, return_data=False):
    rank, world = (0, 1) if return_data else _init_distributed()
    if rank != 0:  # one banner / one log / one set of tables
        logger.setLevel(logging.ERROR)
    print(r"""
    _ _
  /  \ '\                       
  |   \  \      _ _ _  _ _ ___  
   \   \  |    | '_| || | ' \   
    `.__|/     |_|  \_,_|_||_|  [crispr-bean_amd / MI355X]
    """)
    print("bean-run: Run model to identify targeted variants and their impact.")
    bdata = read_h5ad(args.bdata_path)
    args, bdata = check_args(args, bdata)
    prefix = args.outdir + "/bean_run_result." + os.path.basename(args.bdata_path).rsplit(".h5ad", 1)[0]
    os.makedirs(prefix, exist_ok=True)
    if rank == 0:  # rank 0 owns the log file and every table; the other ranks only fit their shards
        handler = logging.FileHandler(f"{prefix}/bean_run.log")
        handler.setLevel(logging.INFO)
        logger.addHandler(handler)
    model_label, model, guide = identify_model_guide(args)
    info("Done loading data. Preprocessing...")
    bdata = prepare_bdata(bdata, args, warn, prefix, write_files=(rank == 0))
    is_neg = lambda scr: np.where(scr.guides[args.negctrl_col].map(lambda v: str(v).lower())  # noqa: E731
                                  == args.negctrl_col_value.lower())[0]
    negctrl_idx = is_neg(bdata) if args.negctrl_col in bdata.guides.columns else np.zeros(0, dtype=int)
    ndata = DATACLASS_DICT[args.selection][model_label](
        bdata,
        repguide_mask=args.repguide_mask,
        sample_mask_column=args.sample_mask_col,
        accessibility_col=args.acc_col,
        accessibility_bw_path=args.acc_bw_path,
        condition_column=args.condition_col,
        time_column=args.time_col,
        control_condition=args.control_condition,
        lower_quantile_column=args.sorting_bin_lower_quantile_col,
        upper_quantile_column=args.sorting_bin_upper_quantile_col,
        target_col=args.target_col,
        shrink_alpha=args.shrink_alpha,
        popt=args.popt,
        use_bcmatch=(not args.ignore_bcmatch),
        negctrl_guide_idx=negctrl_idx,
        allele_df_key=args.allele_df_key,
        control_guide_tag=args.control_guide_tag,
    )
    if args.save_raw and rank == 0:
        pkl.dump(bdata, open(f"{prefix}/ndata.pkl", "wb"))
    if return_data:
        return ndata
    adj_negctrl_idx = None
    control = args.control_condition.split(",")[0]
    if args.library_design == "variant":
        if not args.uniform_edit and "edit_rate" not in ndata.screen.guides.columns:
            ndata.screen.get_guide_edit_rate(unsorted_condition_label=control, condition_col=args.condition_col)
        target_info_df = _get_guide_target_info(ndata.screen, args, cols_include=[args.negctrl_col])
        if args.adjust_confidence_by_negative_control:
            adj_negctrl_idx = np.where(target_info_df[args.negctrl_col].map(lambda v: str(v).lower())
                                       == args.negctrl_col_value.lower())[0]
    else:
        # tiling: one row per edit (bean/cli/run.py:155-206)
        if "edit_rate_norm" not in ndata.screen.guides.columns and "edits" in ndata.screen.layers:
            ndata.screen.get_guide_edit_rate(unsorted_condition_label=control, condition_col=args.condition_col)
        splice = None
        if getattr(args, "splice_site_path", None) is not None:
            import pandas as pd
            splice = pd.read_csv(args.splice_site_path).pos
        target_info_df = variant_table(ndata, ndata.screen.guides.index.values,
                                       bdata.guides["target_group"].values, control_tag=args.control_guide_tag,
                                       splice_sites=splice)
        if args.adjust_confidence_by_negative_control:
            adj_negctrl_idx = np.where((target_info_df.ref == target_info_df.alt)
                                       & (target_info_df.coding == "coding"))[0]
            info(f"Using {len(adj_negctrl_idx)} synonymous variants to adjust confidence.")
    guide_info_df = _get_guide_info(ndata.screen, args, guide_lfc_pseudocount=args.guide_lfc_pseudocount)
    if args.library_design == "tiling":
        import pandas as pd
        guide_info_df = pd.concat([guide_info_df, guide_to_variant_df(target_info_df).reindex(guide_info_df.index)],
                                  axis=1)
    if args.prior_params is not None:
        model = partial(model, prior_params=_check_prior_params(args.prior_params, ndata))

    info(f"Running inference for {model_label}...")
    save_dict = dict()
    param_history_dict_negctrl = None
    if args.load_existing:
        # re-write the tables from a `--save-raw` pickle instead of fitting.  (The reference's branch,
        # bean/cli/run.py:229-231, indexes the loaded dict as if it were the parameter store and
        # fails; here the stored parameters are used.)
        from ..model.run import ParamStore

        with open(f"{prefix}/{model_label}.result{args.result_suffix}.pkl", "rb") as handle:
            loaded = pkl.load(handle)
        param_history_dict = ParamStore(loaded["params"])
        if "negctrl" in loaded:
            param_history_dict_negctrl = ParamStore(loaded["negctrl"]["params"])
        save_dict = loaded
    elif args.fit_negctrl:
        negctrl_model, negctrl_guide = identify_negctrl_model_guide(args, "X_bcmatch" in bdata.layers)
        idx = is_neg(ndata.screen)
        info(f"Using {len(idx)} negative control elements to adjust phenotypic effect sizes...")
        ndata_negctrl = ndata[idx]
        param_history_dict_negctrl, save_dict["negctrl"] = deepcopy(
            run_inference(negctrl_model, negctrl_guide, ndata_negctrl, num_steps=args.n_iter))
        if args.selection == "survival":
            model = partial(model, mu_negctrl=(param_history_dict_negctrl["mu_loc"].detach().mean(),
                                               param_history_dict_negctrl["mu_scale"].detach().mean()))
    if not args.load_existing:
        save_dict["data"] = ndata
        param_history_dict, save_dict_model = deepcopy(run_inference(model, guide, ndata, num_steps=args.n_iter))
        save_dict.update(save_dict_model)
    if rank != 0:
        return prefix
    outfile = f"{prefix}/bean_element[sgRNA]_result.{model_label}{args.result_suffix}.csv"
    info(f"Done running inference. Writing result at {outfile}...")
    if args.save_raw:
        with open(f"{prefix}/{model_label}.result{args.result_suffix}.pkl", "wb") as handle:
            pkl.dump(save_dict, handle)
    write_result_table(
        target_info_df,
        guide_info_df,
        param_history_dict,
        negctrl_params=param_history_dict_negctrl,
        model_label=model_label,
        prefix=f"{prefix}/",
        suffix=args.result_suffix,
        guide_acc=(ndata.guide_accessibility.cpu().numpy() if ndata.guide_accessibility is not None else None),
        adjust_confidence_by_negative_control=args.adjust_confidence_by_negative_control,
        adjust_confidence_negatives=adj_negctrl_idx,
        sd_is_fitted=(args.selection == "sorting"),
        sample_covariates=getattr(ndata, "sample_covariates", None),
        is_survival_screen=(args.selection == "survival"),
    )
    info("Done!")
    return prefix
