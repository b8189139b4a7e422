#!/usr/bin/env python3
"""State-dict layout (keys, shapes, dtypes) of the reference's two networks, from the REAL reference (build
container only): what a `latest.pth` written by src/utils/net_utils.py:325-340 holds under 'net'.
One config per process (the reference's cfg is a global built at import):
    python tools/gen_golden_layout.py nerf && python tools/gen_golden_layout.py kilonerf
Each run merges its part into tests/golden/state_dict_layout.json."""
import json
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
which = sys.argv[1]
sys.dont_write_bytecode = True
os.chdir(REF)
sys.path.insert(0, REF)
sys.argv = ["gen_golden_layout", "--cfg_file", f"configs/{which}/lego.yaml"]
sys.modules["nerfacc"] = types.ModuleType("nerfacc")

from src.config import cfg  # noqa: E402,F401

if which == "nerf":
    from src.models.nerf.network import Network  # noqa: E402
else:
    from src.models.kilonerf.network import Network  # noqa: E402
net = Network()
layout = {k: {"shape": list(v.shape), "dtype": str(v.dtype)} for k, v in net.state_dict().items()}
path = os.path.join(REPO, "tests", "golden", "state_dict_layout.json")
data = json.load(open(path)) if os.path.exists(path) else {}
data[which] = layout
with open(path, "w") as f:
    json.dump(data, f, indent=1, sort_keys=True)
print(which, len(layout), "entries")
