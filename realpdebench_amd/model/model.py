"""``Model`` protocol of the reference (realpdebench/model/model.py:4-26), kept verbatim in meaning:
``forward``, ``train_loss`` and ``load_checkpoint(checkpoint_path, device) -> meta dict``."""
import torch
import torch.nn as nn


class Model(nn.Module):
    def forward(self, x):
        raise NotImplementedError

    def train_loss(self, input, target):
        raise NotImplementedError

    def load_checkpoint(self, checkpoint_path, device):
        """Loads ``checkpoint['model_state_dict']`` (reference key names / dtypes) and returns the same
        meta-data dict as the reference (model.py:14-26)."""
        checkpoint = torch.load(checkpoint_path, map_location=device)
        self.load_state_dict(checkpoint["model_state_dict"])
        return {
            "all_train_losses": checkpoint["train_losses"],
            "all_val_losses": checkpoint["val_losses"],
            "iteration": checkpoint["iteration"],
            "best_iteration": checkpoint["best_iteration"],
            "best_val_loss": checkpoint["best_val_loss"],
        }
