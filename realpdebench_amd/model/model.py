"""The ``Model`` plug-in protocol this backend implements (mirrors the reference's base class, realpdebench/model/model.py:4-26,
in meaning, not in text): ``forward``, ``train_loss`` and ``load_checkpoint(checkpoint_path, device) -> meta dict``."""
import torch
import torch.nn as nn

# checkpoint key -> meta-data key handed back to the caller (train.py:299-301 uses the returned dict)
_META_KEYS = {
    "train_losses": "all_train_losses",
    "val_losses": "all_val_losses",
    "iteration": "iteration",
    "best_iteration": "best_iteration",
    "best_val_loss": "best_val_loss",
}


class Model(nn.Module):
    """Every MI355X model subclasses this; subclasses provide the HIP-backed ``forward`` / ``train_loss``."""

    def forward(self, x):
        raise NotImplementedError(f"{type(self).__name__}.forward")

    def train_loss(self, input, target):
        raise NotImplementedError(f"{type(self).__name__}.train_loss")

    def load_checkpoint(self, checkpoint_path, device):
        """Restores ``model_state_dict`` (reference key names / dtypes; weights only, as in the reference) and returns
        the bookkeeping entries of the checkpoint under the reference's meta-data names."""
        ckpt = torch.load(checkpoint_path, map_location=device)
        self.load_state_dict(ckpt["model_state_dict"])
        return {meta: ckpt[key] for key, meta in _META_KEYS.items()}
