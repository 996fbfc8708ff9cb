"""Seeded inputs of the BASELINE config-3 parity step (3x512 BiLSTM, 425 -> 187), shared by
tests/golden/make_golden.py --config3 (runs the imported reference on them) and
tests/test_gpu_rnn_config3.py (runs the HIP stack on them).  Nothing here is reference code: the
weights are 16 M seeded numbers, far too many to commit, so both sides regenerate them."""
import numpy as np

MODEL_TYPE = "RNNDYN-3_BiLSTM_512-1_FC_187"
IN_DIM, OUT_DIM, HIDDEN = 425, 187, 512
LENGTHS = [48, 48, 47, 45, 44, 41, 40, 40, 39, 37, 33, 32, 31, 30, 29, 27, 25, 24, 23, 21, 20,
           17, 16, 15, 13, 11, 9, 8, 5, 3, 2, 1, 1]          # B = 33: three 16-row batch tiles
N_SAMPLES = 48


def state(shapes, seed=512):
    """shapes: {state-dict key: shape}; values uniform(-1/sqrt(H), 1/sqrt(H)) in sorted key order."""
    rng = np.random.default_rng(seed)
    k = 1.0 / np.sqrt(HIDDEN)
    return {name: rng.uniform(-k, k, size=shapes[name]).astype(np.float32)
            for name in sorted(shapes)}


def batch(seed=33):
    """time-major padded batch in the CALLER's (unsorted) row order: the lengths are shuffled."""
    rng = np.random.default_rng(seed)
    lens = np.array(LENGTHS)[rng.permutation(len(LENGTHS))]
    T, B = int(lens.max()), len(lens)
    x = rng.normal(size=(T, B, IN_DIM)).astype(np.float32)
    y = rng.normal(size=(T, B, OUT_DIM)).astype(np.float32)
    for b, l in enumerate(lens):
        x[l:, b] = 0
        y[l:, b] = 0
    return x, y, lens.astype(np.int64)


def sample_index(name, numel):
    """fixed flat positions at which gradients / updated parameters are stored"""
    rng = np.random.default_rng(abs(hash_name(name)) % (2 ** 32))
    return rng.integers(0, numel, size=min(N_SAMPLES, numel))


def hash_name(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h
