"""Native file I/O of the feature-extraction loop (csrc/hostio.cpp; host code, runs without a
GPU): the batch wav reader equals AudioProcessing.get_raw bit for bit, the batch archive writer
produces what _save_to_npz / np.savez produce as far as np.load and the reader can tell."""
import os

import numpy as np
from scipy.io import wavfile


def test_native_wav_reader_equals_get_raw(golden_dir, tmp_path):
    from idiaptts_amd.src.data_preparation.audio.AudioProcessing import AudioProcessing
    rng = np.random.default_rng(0)
    names = [os.path.join(golden_dir, n + ".wav") for n in ("LJ001-0002", "LJ001-0008")]
    for i, (dtype, scale) in enumerate([(np.int16, 20000), (np.int32, 2 ** 30), (np.float32, 0.5),
                                        (np.uint8, None), (np.float64, 0.5)]):
        x = rng.normal(size=3000 + 17 * i)
        if dtype == np.uint8:
            data = np.clip(x * 40 + 128, 0, 255).astype(np.uint8)
        else:
            data = (np.clip(x / 4, -1, 1) * scale).astype(dtype)
        p = str(tmp_path / "f{}.wav".format(i))
        wavfile.write(p, 22050, data)
        names.append(p)
    stereo = str(tmp_path / "stereo.wav")              # not taken natively: read by get_raw
    wavfile.write(stereo, 16000, (rng.normal(size=(500, 2)) * 3000).astype(np.int16))
    for pre in (0.0, 0.97):
        out, off, rates = AudioProcessing.get_raw_batch(names, pre, n_threads=3)
        assert rates == [16000, 16000] + [22050] * 5
        for i, n in enumerate(names):
            raw, fs = AudioProcessing.get_raw(n, pre)
            assert fs == rates[i]
            assert np.array_equal(out[off[i]:off[i + 1]], raw), n
    mixed = [names[0], stereo, names[3]]
    try:
        out, off, rates = AudioProcessing.get_raw_batch(mixed, 0.0)
    except ValueError:
        out = None      # get_raw itself refuses 2-D audio (np.append flattens: lengths differ)
    if out is not None:
        assert np.array_equal(out[off[2]:off[3]], AudioProcessing.get_raw(names[3], 0.0)[0])


def test_native_archives_read_back_like_np_savez(tmp_path):
    from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import (
        WorldFeatLabelGen, _write_archives_native)
    rng = np.random.default_rng(1)
    for add_deltas, n_bap in ((True, 1), (False, 1), (True, 5)):
        out_dir = str(tmp_path / "d{}{}".format(int(add_deltas), n_bap))
        gen = WorldFeatLabelGen(out_dir, add_deltas=add_deltas, num_coded_sps=20, num_bap=n_bap)
        gen._create_norm_params_extractors()
        cols = gen._cmp_columns()
        width = cols["bap"][0] + cols["bap"][1]
        lens = [31, 1, 250, 2]
        f_off = np.concatenate([[0], np.cumsum(lens)]).tolist()
        cmp_host = rng.normal(size=(f_off[-1], width + 3)).astype(np.float32)[:, :width]  # strided
        names = ["a", "b", "sub/c", "d"]
        streams = [(d, ext, cols[k]) for (_, d, ext, _), k
                   in zip(gen._streams(), ("sp", "lf0", "vuv", "bap"))]
        for d, _, _ in streams:
            os.makedirs(os.path.join(out_dir, d), exist_ok=True)
        assert _write_archives_native(cmp_host, f_off, names, out_dir, streams, add_deltas, 3) == []
        # again: the archives exist with exactly these keys -> replaced, nothing to merge
        assert _write_archives_native(cmp_host, f_off, names, out_dir, streams, add_deltas, 2) == []
        # an archive with a foreign key is left alone and reported
        from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import _save_to_npz
        foreign = os.path.join(out_dir, streams[1][0], "a")
        _save_to_npz(foreign, np.arange(3.0), "something_else")
        before = open(foreign + ".npz", "rb").read()
        assert _write_archives_native(cmp_host, f_off, names, out_dir, streams, add_deltas, 2) \
            == [(0, 1)]
        assert open(foreign + ".npz", "rb").read() == before
        os.remove(foreign + ".npz")
        assert _write_archives_native(cmp_host, f_off, names, out_dir, streams, add_deltas, 2) == []
        ref_dir = out_dir + "_ref"
        ref = WorldFeatLabelGen(ref_dir, add_deltas=add_deltas, num_coded_sps=20, num_bap=n_bap)
        ref._create_norm_params_extractors()
        for u, n in enumerate(names):
            ref._write_utterance(ref_dir, os.path.basename(n), cmp_host[f_off[u]:f_off[u + 1]], cols)
            got = gen.load(os.path.basename(n))
            assert np.array_equal(got, cmp_host[f_off[u]:f_off[u + 1]])
            assert np.array_equal(got, ref.load(os.path.basename(n)))
            for d, ext, _ in streams:
                import zipfile
                with zipfile.ZipFile(os.path.join(out_dir, d, os.path.basename(n) + ".npz")) as zf:
                    assert zf.testzip() is None              # every member's CRC-32 (taken row by row while writing)
                a = np.load(os.path.join(out_dir, d, os.path.basename(n) + ".npz"))
                b = np.load(os.path.join(ref_dir, d, os.path.basename(n) + ".npz"))
                assert sorted(a.files) == sorted(b.files)
                for k in a.files:
                    assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k])
