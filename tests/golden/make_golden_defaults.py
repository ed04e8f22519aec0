"""Constructor defaults of the reference's five HF dataset classes, read by IMPORTING them (build container only):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_defaults.py
Writes tests/golden/scenario_defaults.json -- data only: {scenario: {in_step, out_step, n_sim_frame, sub_s_real, sub_s_numerical,
mask_prob}}.  realpdebench/train.py:118-266 passes none of these, so they define the benchmark's windows per scenario."""
import inspect
import json
import os
import sys
import types

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
for name in ("h5py",):                         # import-only dependency of the legacy readers
    sys.modules.setdefault(name, types.ModuleType(name))
from realpdebench.data.combustion_hf_dataset import CombustionHFDataset                      # noqa: E402
from realpdebench.data.fluid_hf_dataset import (ControlledCylinderHFDataset, CylinderHFDataset, FoilHFDataset,   # noqa: E402
                                                FSIHFDataset)

KEYS = ("in_step", "out_step", "n_sim_frame", "sub_s_real", "sub_s_numerical", "mask_prob")
CLASSES = {"cylinder": CylinderHFDataset, "fsi": FSIHFDataset, "controlled_cylinder": ControlledCylinderHFDataset,
           "foil": FoilHFDataset, "combustion": CombustionHFDataset}
out = {}
for scen, cls in CLASSES.items():
    sig = inspect.signature(cls.__init__)
    out[scen] = {k: sig.parameters[k].default for k in KEYS}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenario_defaults.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print(json.dumps(out))
