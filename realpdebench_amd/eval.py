"""``python -m realpdebench_amd.eval --config ... --checkpoint_path model_XXXX.pth`` -- the reference's evaluation
entrypoint (realpdebench/eval.py): load a checkpoint, roll the model out autoregressively (eval.py:311-321) and report
the normalised MSE plus RMSE / MAE / Rel-L2 in physical units and the full ``eval_metrics`` tuple of
utils/metrics.py:24-131 (R^2, kinetic energy, radially binned Fourier errors, frequency error), computed on the device by
``realpdebench_amd.metrics`` (truncated-DFT GEMMs instead of fftn + a Python triple loop)."""
import argparse
import logging
import os
import time

import torch
from torch.utils.data import DataLoader

from .data import make_datasets
from .data_normalizer import GaussianNormalizer, IdentityNormalizer
from .metrics import eval_metrics
from .model import load_model
from .rollout import autoregressive_rollout
from .utils import add_args_from_config, resolve_config, set_seed, setup_logging

parser = argparse.ArgumentParser(description="Evaluation Configurations")
parser.add_argument("--config", type=str, default="configs/cylinder/fno.yaml")
parser.add_argument("--gpu", type=int, default=0)
parser.add_argument("--checkpoint_path", type=str, default=None)
parser.add_argument("--dataset_factory", type=str, default=None)


def main(argv=None):
    args = parser.parse_args(argv)
    args.config = resolve_config(args.config)
    cli_ckpt = args.checkpoint_path
    args = add_args_from_config(args)
    if cli_ckpt:
        args.checkpoint_path = cli_ckpt
    if not torch.cuda.is_available():
        raise SystemExit("realpdebench_amd.eval needs an MI355X: there is no CPU fallback path")
    device = torch.device("cuda", args.gpu)
    torch.cuda.set_device(device)
    set_seed(args.seed)
    exp_path = os.path.dirname(args.checkpoint_path) if args.checkpoint_path else args.results_path
    os.makedirs(exp_path, exist_ok=True)
    setup_logging(exp_path, is_train=False)

    train_dataset, test_dataset, stats = make_datasets(args, for_eval=True)
    loader = DataLoader(test_dataset, batch_size=args.test_batch_size, shuffle=False, num_workers=args.num_workers)
    normalizer = GaussianNormalizer(*stats, device=device) if args.normalizer == "gaussian" else IdentityNormalizer(device)
    model = load_model(train_dataset, device=device, **vars(args))
    if args.checkpoint_path:
        meta = model.load_checkpoint(args.checkpoint_path, device)                     # eval.py:284
        logging.info(f"Checkpoint {args.checkpoint_path} loaded (iteration {meta['iteration']}).")
    n_ar = int(args.N_autoregressive)
    se = ae = ref2 = nmse = 0.0
    cnt = nb = 0
    metric_rows = []
    start = time.time()
    for inp, tgt in loader:
        c_out = tgt.shape[-1]
        para = inp[..., c_out:].contiguous() if inp.shape[-1] != c_out else None          # eval.py:305-309
        x, t = normalizer.preprocess(inp, tgt)
        pred = autoregressive_rollout(model, x, n_ar, normalizer=normalizer, para_input=para)[..., :c_out]
        t_roll = t if n_ar == 1 else None
        if t_roll is not None:
            nmse += float(((pred - t_roll) ** 2).mean()) * inp.shape[0]
            _, p = normalizer.postprocess(x, pred.contiguous())
            _, tt = normalizer.postprocess(x, t_roll)
            se += float(((p - tt) ** 2).sum())
            ae += float((p - tt).abs().sum())
            ref2 += float((tt ** 2).sum())
            cnt += tt.numel()
            metric_rows.append(torch.stack(eval_metrics(p, tt, c_out)) * inp.shape[0])     # eval.py:327-333, on device
        nb += inp.shape[0]
    torch.cuda.synchronize()
    dt = time.time() - start
    T_out = tuple(test_dataset[0][1].shape)[0]
    logging.info(f"rollout: {nb} trajectories x {n_ar} steps in {dt:.2f} s = {nb * T_out * n_ar / dt:.1f} fields/s")
    if cnt:
        logging.info(f"normalized mse {nmse / nb:.5f}, rmse {(se / cnt) ** 0.5:.5f}, mae {ae / cnt:.5f}, "
                     f"rel l2 {(se / max(ref2, 1e-30)) ** 0.5:.5f}")
        names = ("rmse", "mae", "rel_l2_error", "r2", "ke_error", "f_error", "low_f_error", "mid_f_error", "high_f_error",
                 "rel_low_f_error", "rel_mid_f_error", "rel_high_f_error", "freq_error")
        avg = torch.stack(metric_rows).sum(0) / nb
        logging.info("eval_metrics (batch-weighted mean): " + ", ".join(f"{n} {float(v):.5g}" for n, v in zip(names, avg)))


if __name__ == "__main__":
    main()
