"""``python -m realpdebench_amd.train --config configs/cylinder/fno.yaml`` -- the reference's training entrypoint
(realpdebench/train.py) on the MI355X backend: same flags, same YAML merge, same checkpoint format.

Differences, all on the hot path (train.py:321-342): the step is the fused ``Trainer.step`` (no per-step ``.item()``
syncs; the loss is read back once per logging interval) and, when launched under ``torch.distributed.run`` with
several processes, batches are sharded across GPUs and gradients reduced with RCCL (``dp.DataParallel``)."""
import argparse
import datetime
import logging
import os
import time

import torch
from torch.utils.data import DataLoader

from .data import DevicePrefetcher, make_datasets
from .data_normalizer import GaussianNormalizer, IdentityNormalizer
from .model import load_model
from .trainer import make_trainer
from .utils import add_hf_compat_flags, add_args_from_config, check_hf_compat_flags, cycle, resolve_config, set_seed, setup_logging

parser = argparse.ArgumentParser(description="Training Configurations")
parser.add_argument("--config", type=str, default="configs/cylinder/fno.yaml")
parser.add_argument("--gpu", type=int, default=0)
parser.add_argument("--train_data_type", type=str, default="numerical", help="numerical | real")
parser.add_argument("--is_finetune", action="store_true", help="enable finetuning mode")
parser.add_argument("--dataset_factory", type=str, default=None, help="module:function -> (train, val, stats)")
parser.add_argument("--max_updates", type=int, default=None, help="stop early (smoke runs); the schedule still uses num_update")
add_hf_compat_flags(parser)


VAL_KEYS = ("rmse", "mae", "rel_l2_error", "r2", "ke_error", "f_error", "low_f_error", "mid_f_error", "high_f_error",
            "rel_low_f_error", "rel_mid_f_error", "rel_high_f_error", "freq_error")


@torch.no_grad()
def validate(model, val_loader, normalizer):
    """The validation pass of realpdebench/train.py:344-373: one forward per batch, the normalised MSE over the measured channels
    (all-zero target channels of the first batch are skipped: real data has no pressure) as the mean of per-batch means, and
    ``eval_metrics`` once over the de-normalised, concatenated split -- on the device (``realpdebench_amd.metrics``)."""
    from .metrics import eval_metrics
    model.eval()
    nmse, preds, tgts, c = 0.0, [], [], None
    for vi, vt in val_loader:
        b = vi.size(0)
        if c is None:
            c = vt.shape[-1] - sum(int(torch.all(vt[..., k] == 0)) for k in range(vt.shape[-1]))
        vi, vt = normalizer.preprocess(vi, vt)
        pred = model(vi)
        nmse += float(((pred[..., :c] - vt[..., :c]) ** 2).reshape(b, -1).mean())
        _, p = normalizer.postprocess(vi, pred)
        _, t = normalizer.postprocess(vi, vt)
        preds.append(p)
        tgts.append(t)
    return nmse / len(val_loader), eval_metrics(torch.cat(preds), torch.cat(tgts), c)


def main(argv=None):
    args = parser.parse_args(argv)
    check_hf_compat_flags(args)
    args.config = resolve_config(args.config)
    args = add_args_from_config(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(args.gpu)))
    if not torch.cuda.is_available():
        raise SystemExit("realpdebench_amd.train needs an MI355X: there is no CPU fallback path")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    set_seed(args.seed)

    exp_path = os.path.join(args.results_path, args.model_name,
                            f"{args.exp_name}_{args.train_data_type}_{args.is_finetune}",
                            datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S"))
    if rank == 0:
        os.makedirs(exp_path, exist_ok=True)
        setup_logging(exp_path)
        logging.info(f"args: {args}")

    train_dataset, val_dataset, stats = make_datasets(args)
    if args.train_batch_size % world != 0 or args.train_batch_size < world:
        raise ValueError(f"train_batch_size={args.train_batch_size} must be a positive multiple of the {world} data-parallel ranks")
    per_rank_bs = args.train_batch_size // world
    if len(train_dataset) < args.train_batch_size:
        raise ValueError(f"the training set has {len(train_dataset)} samples, fewer than one global batch of "
                         f"{args.train_batch_size}: with drop_last the loader would never yield a batch")
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(train_dataset, num_replicas=world, rank=rank, shuffle=True, seed=args.seed)
    from .disk import DiskBatchLoader, FluidWindows
    on_disk = isinstance(train_dataset, FluidWindows)
    train_loader = None if on_disk else cycle(DataLoader(train_dataset, batch_size=per_rank_bs, shuffle=sampler is None,
                                                         sampler=sampler, pin_memory=True, num_workers=args.num_workers,
                                                         drop_last=True), sampler=sampler)   # new permutation every epoch
    val_loader = DataLoader(val_dataset, batch_size=args.test_batch_size, shuffle=False, num_workers=args.num_workers)
    if args.normalizer == "gaussian":
        if stats is None:
            raise ValueError("normalizer: gaussian needs (mean_in, mean_tgt, std_in, std_tgt) from the dataset factory")
        normalizer = GaussianNormalizer(*stats, device=device)
    elif args.normalizer == "none":
        normalizer = IdentityNormalizer(device)
    else:
        raise ValueError(f"Normalizer {args.normalizer} not supported")

    model = load_model(train_dataset, device=device, **vars(args))
    if args.is_finetune:
        model.load_checkpoint(args.checkpoint_path, device)
        logging.info(f"Checkpoint {args.checkpoint_path} loaded.")
    trainer = make_trainer(model, lr=args.lr, num_update=args.num_update, scheduler=args.scheduler,
                           step_size=args.step_size, clip_grad_norm=args.clip_grad_norm,   # wraps DP when world > 1
                           micro_batch=getattr(args, "micro_batch_size", None))           # optional YAML key (not in the reference)

    n_iter = args.num_update if args.max_updates is None else min(args.num_update, args.max_updates)
    every = max(1, int(args.num_update / 50))                       # train.py:344
    all_train_losses, all_val_losses = [], {k: [] for k in ("normalized_mse",) + VAL_KEYS}        # train.py:303-319
    best_val, best_it = float("inf"), 0
    pending, start = [], time.time()
    if on_disk:       # row f2: memory-mapped Arrow slabs -> pinned staging -> rpb_window_pack (+ normaliser) on a side stream
        batches = DiskBatchLoader(train_dataset, per_rank_bs, device, stats=stats if args.normalizer == "gaussian" else None,
                                  shuffle=True, seed=args.seed, rank=rank, world=world)
    else:
        batches = DevicePrefetcher(train_loader, normalizer, device)     # async H2D + normalise on a side stream (row f1)
    for iteration in range(1, n_iter + 1):
        inp, tgt = next(batches)
        pending.append(trainer.step(inp, tgt).clone())              # device scalar, no sync
        if iteration % every == 0 or iteration == n_iter:
            all_train_losses += [float(v) for v in torch.cat(pending).cpu()]    # ONE sync per interval
            pending = []
            nmse, vals = validate(model, val_loader, normalizer)
            all_val_losses["normalized_mse"].append(nmse)
            for k, v in zip(VAL_KEYS, vals):
                all_val_losses[k].append(float(v))
            rmse = float(vals[0])
            if rmse < best_val:
                best_val, best_it = rmse, iteration
            if rank == 0:
                logging.info(f"\nIteration {iteration}, train loss: {sum(all_train_losses[-every:]) / every:.5f} "
                             f"(lr {trainer.current_lr():.3e})")
                logging.info("Validation results: \n" + f"normalized mse loss: {nmse:.5f}, "
                             + ", ".join(f"{k.replace('_', ' ')}: {float(v):.5f}" for k, v in zip(VAL_KEYS, vals)))
                torch.save({"model_state_dict": model.state_dict(), "train_losses": all_train_losses,
                            "val_losses": all_val_losses, "iteration": iteration, "best_iteration": best_it,
                            "best_val_loss": best_val}, os.path.join(exp_path, f"model_{iteration:04d}.pth"))
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.time() - start
        logging.info(f"Training complete, best iteration is {best_it}, time cost is {dt / 60:.2f} min "
                     f"({n_iter * args.train_batch_size / dt:.1f} samples/s incl. validation)")
        logging.info(f"Results saved at {exp_path}")
    if world > 1:
        torch.distributed.destroy_process_group()
    return exp_path


if __name__ == "__main__":
    main()
