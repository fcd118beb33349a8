"""Diagnostic (GPU box): tests/test_edge_batches.py::_check over several ray counts and seeds; prints which (R, seed) trip the bulk rule."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import test_edge_batches as T
orig = T._batch
for R in (33, 34, 40, 47):
    for off in (100, 200, 300, 400):
        T._batch = lambda R_, seed, off=off: orig(R_, off + R_)
        try:
            T._check(R, None, torch.device("cuda:0"), "dtu"); print(R, off, "ok")
        except AssertionError as e:
            print(R, off, "FAIL", str(e)[:160])
