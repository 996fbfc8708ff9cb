import os, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.multiprocessing as mp
import test_gpu_dp as T
if __name__ == "__main__":
    cache = sys.argv[1] == "1"
    root = tempfile.mkdtemp()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(T._trainer_dp_worker, args=(2, T._free_port(), ret, root, cache), nprocs=2, join=True)
    print({r: (ret[r][0], ret[r][1], ret[r][3], ret[r][4]) for r in (0, 1)})
