"""A/A noise of the tiny full-objective training run (two eager runs from the same seed) next to graph-vs-eager: tells a real
replay bug from atomics noise amplified by Adam + the adaptive GAN weight.  GPU box only."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_stepgraph as T  # noqa: E402

dev = torch.device("cuda:0")
for loss in ("ae", "full"):
    _, _, e1 = T._run(dev, False, 8, loss)
    _, _, e2 = T._run(dev, False, 8, loss)
    _, _, g1 = T._run(dev, True, 8, loss)
    _, _, g2 = T._run(dev, True, 8, loss)
    rel = lambda a, b: np.abs(a - b) / (np.abs(b) + 1e-9)
    print(loss, "eager-eager max rel per step", rel(e1, e2).max(axis=1).round(5).tolist())
    print(loss, "graph-eager max rel per step", rel(g1, e1).max(axis=1).round(5).tolist())
    print(loss, "graph-graph max rel per step", rel(g1, g2).max(axis=1).round(5).tolist())
