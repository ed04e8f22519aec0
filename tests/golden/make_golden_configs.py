"""Hyper-parameters of every in-scope reference YAML, read from /root/reference (build container only):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_configs.py
Writes tests/golden/reference_configs.json -- data only:
    {"native_shapes": {scenario: {"shape_in": [T,H,W,C_in], "shape_out": [T,H,W,C_out]}},
     "configs": {scenario: {yaml stem: {key: value, ...}}}}
for the five scenarios x the four north-star models (fno, unet, trainsolver (sic), galerkin_transformer) + dpot_s / dpot_l.
The values are `yaml.safe_load` of realpdebench/configs/<scenario>/<stem>.yaml, nothing else.

`native_shapes` is what `train_dataset[0]` hands `load_model` (model/load_model.py:7-9) for numerical data at each dataset
class's default windows (tests/golden/scenario_defaults.json) -- the data files are not in the container, so the mesh sizes are
SURVEY.md section 8's (fluid_hf_dataset.py:258-302,374-375,404-552; combustion_hf_dataset.py:264-349) and this script cross-checks
every one of them against the reference's own trainsolver YAML, whose H*W*D must equal T*H*W and whose space_dim / out_dim
must equal C_in / C_out for the model to run at all (Transolver_Structured_Mesh_3D.py:170-196).
`--write-yamls` re-emits realpdebench_amd/configs/<scenario>/<stem>.yaml from the same values (synthetic dataset defaults on
top, see the header each file carries)."""
import json
import os
import sys

import yaml

REF = "/root/reference/realpdebench/configs"
HERE = os.path.dirname(os.path.abspath(__file__))
SCENARIOS = ("cylinder", "controlled_cylinder", "fsi", "foil", "combustion")
STEMS = ("fno", "unet", "trainsolver", "galerkin_transformer", "dpot_s", "dpot_l")
NATIVE = {
    "cylinder": ((20, 64, 128, 3), (20, 64, 128, 3)),
    "controlled_cylinder": ((10, 64, 128, 5), (10, 64, 128, 3)),      # two control channels in, three fields out
    "fsi": ((20, 64, 64, 3), (20, 64, 64, 3)),
    "foil": ((20, 64, 128, 3), (20, 64, 128, 3)),
    "combustion": ((20, 64, 64, 16), (20, 64, 64, 16)),               # observed + 15 numerical channels
}


def main():
    configs = {}
    for scen in SCENARIOS:
        configs[scen] = {}
        for stem in STEMS:
            with open(os.path.join(REF, scen, stem + ".yaml")) as fh:
                configs[scen][stem] = yaml.safe_load(fh)
        ts = configs[scen]["trainsolver"]
        (T, H, W, Ci), (To, _, _, Co) = NATIVE[scen]
        assert ts["H"] * ts["W"] * ts["D"] == T * H * W, (scen, ts["H"], ts["W"], ts["D"])
        assert ts["space_dim"] == Ci and ts["out_dim"] == Co and ts["D"] == T, scen
    out = {"native_shapes": {s: {"shape_in": list(NATIVE[s][0]), "shape_out": list(NATIVE[s][1])} for s in SCENARIOS},
           "configs": configs}
    with open(os.path.join(HERE, "reference_configs.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote reference_configs.json:", sum(len(v) for v in configs.values()), "configs")
    if "--write-yamls" in sys.argv:
        write_yamls(out)


# keys the shipped YAMLs override so that `python -m realpdebench_amd.train --config ...` runs without a dataset on disk
SYNTH = ("dataset_name", "dataset_root", "num_workers", "normalizer", "checkpoint_path")


def write_yamls(doc):
    root = os.path.join(os.path.dirname(os.path.dirname(HERE)), "realpdebench_amd", "configs")
    for scen, per in doc["configs"].items():
        os.makedirs(os.path.join(root, scen), exist_ok=True)
        shp = doc["native_shapes"][scen]
        for stem, cfg in per.items():
            path = os.path.join(root, scen, stem + ".yaml")
            if os.path.exists(path) and "--overwrite" not in sys.argv:
                continue                        # hand-written files of earlier rounds (cylinder/*, fsi/fno) stay as they are
            head = (f"# Key surface and values of the reference's realpdebench/configs/{scen}/{stem}.yaml "
                    "(written by tests/golden/make_golden_configs.py --write-yamls).\n"
                    "# Deviations, on purpose: dataset_name / dataset_root default to the synthetic generator (no dataset ships "
                    "here), normalizer to \"none\"\n# (synthetic fields are N(0,1)) and checkpoint_path to \"\"; the reference's "
                    "values are kept in the ref_* keys below -- set them back together\n# with a real dataset_root.\n")
            body = dict(cfg)
            for k in SYNTH:
                if k in body:
                    body["ref_" + k] = body[k]
            body.update(dataset_name="synthetic", dataset_root="", num_workers=0, normalizer="none", checkpoint_path="",
                        shape_in=shp["shape_in"], shape_out=shp["shape_out"], n_train=64, n_val=16)
            if stem.startswith("dpot"):         # the YAML's path names the pretrained weights the reference downloads: none here -> random init
                body["checkpoint_path"] = None
            with open(path, "w") as fh:
                fh.write(head)
                yaml.safe_dump(body, fh, sort_keys=False, default_flow_style=None)
    print("wrote YAMLs under", root)


if __name__ == "__main__":
    main()
