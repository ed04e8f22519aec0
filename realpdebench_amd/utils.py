"""Host glue with the reference's semantics (realpdebench/utils/utils.py): YAML -> args merge, seeding, logging."""
import logging
import os

import numpy as np
import torch
import yaml


def add_args_from_config(args):
    """Every YAML key not already a CLI argument becomes ``args.<key>`` (utils.py:13-22)."""
    existing = set(vars(args).keys())
    with open(args.config, "r") as f:
        config = yaml.safe_load(f)
    for key, value in config.items():
        if key not in existing:
            setattr(args, key, value)
    return args


def resolve_config(path):
    """Falls back to a package-relative path like the reference does (train.py:58-61)."""
    if not os.path.exists(path):
        cand = os.path.join(os.path.dirname(__file__), path)
        if os.path.exists(cand):
            return cand
    return path


def set_seed(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def setup_logging(exp_path, is_train=True):
    log_filename = os.path.join(exp_path, "training.log" if is_train else "eval.log")
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s",
                        handlers=[logging.FileHandler(log_filename), logging.StreamHandler()], force=True)
    logging.info(f"Logging initialized at {log_filename}")


def cycle(iterable, sampler=None):
    """utils/utils.py ``cycle``: restart the loader forever.  ``sampler``: a DistributedSampler whose ``set_epoch`` is called at
    every restart so that each epoch draws a new permutation (without it every epoch reuses the first one); an epoch that yields
    nothing raises instead of spinning."""
    epoch = 0
    while True:
        if sampler is not None:
            sampler.set_epoch(epoch)
        n = 0
        for x in iterable:
            n += 1
            yield x
        if n == 0:
            raise RuntimeError("cycle(): the loader produced no batch in a whole epoch (dataset smaller than one batch with "
                               "drop_last?)")
        epoch += 1


def add_hf_compat_flags(parser):
    """The Arrow-reader switches of the reference's entrypoints (train.py:29-53, eval.py:30-54, train_surrogate.py:20-46).  This
    backend always reads the Arrow layout, so ``--use_hf_dataset`` changes nothing; there is no network path, so
    ``--hf_auto_download`` is refused by ``check_hf_compat_flags`` instead of being silently ignored."""
    if not any(a.dest == "use_hf_dataset" for a in parser._actions):
        parser.add_argument("--use_hf_dataset", action="store_true", help="accepted for CLI compatibility (Arrow is the only reader)")
    parser.add_argument("--hf_auto_download", action="store_true", help="not supported: fetch the data with the reference's CLI")
    parser.add_argument("--hf_repo_id", type=str, default="AI4Science-WestlakeU/RealPDEBench")
    parser.add_argument("--hf_endpoint", type=str, default=None)
    parser.add_argument("--hf_revision", type=str, default=None)


def check_hf_compat_flags(args):
    if getattr(args, "hf_auto_download", False):
        raise SystemExit("--hf_auto_download: this backend has no download path; fetch the dataset with the reference "
                         "(`realpdebench download --dataset-root ... --what hf_dataset`) and point dataset_root at it")
