"""Materialises tests/golden/trainer_fixture.npz (the data files of the reference's trainer
tests, test/integration/fixtures/{questions,WORLD/cmp_mcep20}) in the directory layout the
trainers read: <root>/questions/<id>.questions + min-max.bin, <root>/WORLD/cmp_mcep20/<id>.cmp +
the legacy mean-covariance .bin files."""
import os

import numpy as np


def materialise(golden_dir, root):
    g = np.load(os.path.join(golden_dir, "trainer_fixture.npz"))
    os.makedirs(os.path.join(root, "questions"), exist_ok=True)
    os.makedirs(os.path.join(root, "WORLD", "cmp_mcep20"), exist_ok=True)
    ids = [str(i) for i in g["id_list"]]
    for i in ids:
        g["questions/" + i].astype(np.float32).tofile(
            os.path.join(root, "questions", i + ".questions"))
        g["cmp/" + i].astype(np.float32).tofile(
            os.path.join(root, "WORLD", "cmp_mcep20", i + ".cmp"))
    for k in g.files:
        if k.startswith("bin/"):
            g[k].tofile(os.path.join(root, k[4:]))
    return ids, os.path.join(root, "WORLD"), os.path.join(root, "questions"), g
