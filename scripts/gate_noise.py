"""How noisy are the statistics the EXACT_GATE of tests/test_gpu_amp.py compares?  Two fp32-accumulating evaluations of the SAME
policy on the CPU -- oracle/nets_amp.py in its normal and in its reversed summation order -- each measured against the
exactly-rounded yardstick (tests/golden/acc64_chain_golden.npz) on the 252 hypotheses of teacher-forced iteration 0 (and 1): the
ratio of their median / p90 / p99 / max distances is pure sampling noise of two samples from one distribution.  CPU only.
    python scripts/gate_noise.py > profiles/r04_gate_noise.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from amp_util import geodesic  # noqa: E402
from conftest import _build_scene  # noqa: E402
from foundationpose_amd.weights import DEFAULT_REFINE_CFG, random_state_dict  # noqa: E402
from oracle import nets_amp, ops as oo  # noqa: E402
from oracle import pipeline as op  # noqa: E402

G = dict(np.load(os.path.join(ROOT, "tests", "golden", "acc64_chain_golden.npz")))
sc = _build_scene()
d = op.preprocess_depth(sc["depth"])
xyz = oo.depth2xyzmap(d, sc["K"], f64_internal=True)
cfg = dict(DEFAULT_REFINE_CFG)
sd = random_state_dict("refine", cfg, seed=0)
tn = [float(v) for v in cfg["trans_normalizer"]]


def pct(x):
    return dict(median=float(np.median(x)), p90=float(np.percentile(x, 90)), p99=float(np.percentile(x, 99)), max=float(np.max(x)))


out = {"iterations": []}
pool = {"normal": ([], []), "reversed": ([], [])}
for it in range(int(os.environ.get("FP_ITERS", "2"))):
    start, exact = G["tf_start"][it], G["tf_exact"][it]
    A, B, _, _ = op.refine_inputs(cfg, start, sc["mesh_np"], sc["rgb"], xyz, sc["K"], sc["diameter"])
    row = {}
    for name, rev in (("normal", False), ("reversed", True)):
        nets_amp.REVERSED_SUMS = rev
        try:
            o = nets_amp.refine_forward(torch.from_numpy(A), torch.from_numpy(B), sd)
        finally:
            nets_amp.REVERSED_SUMS = False
        p = oo.pose_update(o["trans"].numpy(), o["rot"].numpy(), start, cfg["rot_rep"], True, tn, float(cfg["rot_normalizer"]), float(sc["diameter"]))
        dR = geodesic(p[:, :3, :3], exact[:, :3, :3])
        dt = np.linalg.norm(p[:, :3, 3].astype(np.float64) - exact[:, :3, 3], axis=1)
        pool[name][0].append(dR); pool[name][1].append(dt)
        row[name + "_to_exact"] = dict(dR=pct(dR), dt=pct(dt))
    row["ratio_reversed_over_normal"] = {q: {s: row["reversed_to_exact"][q][s] / row["normal_to_exact"][q][s] for s in ("median", "p90", "p99", "max")}
                                         for q in ("dR", "dt")}
    out["iterations"].append(row)
pp = {n: dict(dR=pct(np.concatenate(v[0])), dt=pct(np.concatenate(v[1]))) for n, v in pool.items()}
out["pooled"] = pp
out["pooled_ratio_reversed_over_normal"] = {q: {s: pp["reversed"][q][s] / pp["normal"][q][s] for s in ("median", "p90", "p99", "max")} for q in ("dR", "dt")}
print(json.dumps(out, indent=1))
