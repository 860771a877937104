"""Drop-in check data: the flat experiment dicts the reference's own generator emits for BASELINE configs 1-5
(`experiments.generate_experiment_cfgs(40|42|43|44)[0]`, experiments.py:60-102,373-456) and the EVALUATED contents of the
model hyper-parameter files `build_model` reads (configs/_base_/models/*.py -> plain dicts).  Data only (JSON), no
source.  Build container only.  Writes tests/golden/experiment_cfgs.json."""
import contextlib
import io
import json
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    import experiments
    out = dict(experiments={}, model_cfgs={})
    for e in (40, 42, 43, 44):
        with contextlib.redirect_stderr(io.StringIO()):
            out["experiments"][str(e)] = experiments.generate_experiment_cfgs(e)[0]
    for name in ("vlm-vlg-aspp-s2p4-sk04-ftap-mcvitb", "vlm-vlg-aspp-s2p4-skr04-ftap-mcvitb", "mcvit16"):
        ns = runpy.run_path(f"configs/_base_/models/{name}.py")
        out["model_cfgs"][name] = {k: v for k, v in ns.items() if not k.startswith("_")}
    path = os.path.join(HERE, "experiment_cfgs.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True, default=lambda o: list(o))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
