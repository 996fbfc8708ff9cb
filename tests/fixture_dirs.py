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


def materialise_duration(golden_dir, root):
    """Duration-model fixtures (tests/golden/duration_fixture.npz + labels_state_align.zip) as
    <root>/dur/<id>.dur + mean-std_dev.bin, <root>/labels/{label_state_align,mono_no_align}/<id>.lab,
    <root>/labels/mono_phone.list."""
    import zipfile
    g = np.load(os.path.join(golden_dir, "duration_fixture.npz"))
    ids = [str(i) for i in g["id_list"]]
    for sub in ("dur", "labels/label_state_align", "labels/mono_no_align"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    zipfile.ZipFile(os.path.join(golden_dir, "labels_state_align.zip")).extractall(
        os.path.join(root, "labels", "label_state_align"))
    for i in ids:
        g["dur/" + i].astype(np.float32).tofile(os.path.join(root, "dur", i + ".dur"))
        with open(os.path.join(root, "labels", "mono_no_align", i + ".lab"), "w") as f:
            f.write(str(g["mono_no_align/" + i]))
    g["bin/dur/mean-std_dev.bin"].tofile(os.path.join(root, "dur", "mean-std_dev.bin"))
    with open(os.path.join(root, "labels", "mono_phone.list"), "w") as f:
        f.write(str(g["mono_phone_list"]))
    return ids, g
