"""Host-side logic vs vectors captured from the imported reference (tests/golden/make_golden.py).
Bit-exact for indexing / integer / float32 sequencing semantics."""
import os

import numpy as np
import pytest

from idiaptts_amd.misc.utils import compute_deltas, interpolate_lin


@pytest.fixture(scope="module")
def host(golden_dir):
    return np.load(os.path.join(golden_dir, "host_logic.npz"))


def test_interpolate_lin_bit_exact(host):
    for i in range(int(host["il_count"])):
        ip, vuv = interpolate_lin(host["il_in_%d" % i])
        assert ip.dtype == host["il_ip_%d" % i].dtype
        assert np.array_equal(ip, host["il_ip_%d" % i]), i
        assert np.array_equal(vuv, host["il_vuv_%d" % i]), i


def test_compute_deltas_bit_exact(host):
    for i in range(int(host["cd_count"])):
        out = compute_deltas(host["cd_in_%d" % i])
        assert out.dtype == np.float32 and np.array_equal(out, host["cd_out_%d" % i])
    # SURVEY.md Appendix C known answer
    x = np.array([[1, 2], [2, 5], [4, 4], [8, 0]], dtype=np.float32)
    assert np.array_equal(compute_deltas(x), np.array([[1, 3], [1.5, 1], [3, -2.5], [4, -4]],
                                                       dtype=np.float32))


def test_state_align_durations_bit_exact(golden_dir):
    """reference fixtures dur/*.dur == _get_full_state_align_dur(label_state_align/*.lab)
    (float32 parse + float32 division, PhonemeDurationLabelGen.py:306-314)."""
    from idiaptts_amd.src.data_preparation.phonemes.PhonemeDurationLabelGen import \
        PhonemeDurationLabelGen as P
    for name in ["LJ001-0002", "LJ001-0008"]:
        dur = P._get_full_state_align_dur(os.path.join(golden_dir, name + ".lab"), 50000, 5)
        ref = np.fromfile(os.path.join(golden_dir, name + ".dur"), dtype=np.float32).reshape(-1, 5)
        assert dur.dtype == np.float32 and np.array_equal(dur, ref)
        A = P.convert_to_matrix(dur)
        assert A.shape == (int(ref.sum()), ref.shape[0]) and A.dtype == np.float32
        assert np.array_equal(A.sum(0), ref.sum(1))
    # docstring example of the reference (:181-189)
    A = P.durations_to_hard_attention_matrix(np.array([3, 0, 1, 2]))
    assert np.array_equal(A, np.array([[1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0],
                                       [0, 0, 0, 1], [0, 0, 0, 1]], dtype=np.float32))
    assert np.array_equal(P.load_sample("LJ001-0008", golden_dir),
                          np.fromfile(os.path.join(golden_dir, "LJ001-0008.dur"),
                                      dtype=np.float32).reshape(-1, 5))


def test_length_matching_index_math():
    """WORLD 1931 frames vs questions 1926 -> trim front 2, end 3 (SURVEY.md Appendix C;
    PyTorchDatareadersDataset.py:179-197); the longer stream is the one trimmed."""
    from idiaptts_amd.src.data_preparation.DataReaders import match_lengths, trim_to_reference
    w = np.arange(1931 * 2).reshape(1931, 2)
    q = np.arange(1926 * 3).reshape(1926, 3)
    t, was = trim_to_reference(w, [1926])
    assert was and t.shape[0] == 1926 and t[0, 0] == w[2, 0] and t[-1, 0] == w[-4, 0]
    with pytest.raises(ValueError):
        trim_to_reference(q, [1931])
    out = match_lengths({"acoustic_features": w, "questions": q},
                        {"acoustic_features": ["questions"], "questions": ["acoustic_features"]})
    assert out["acoustic_features"].shape[0] == out["questions"].shape[0] == 1926
    assert np.array_equal(out["questions"], q)
    same, was = trim_to_reference(q, [1926])
    assert not was and same is q


def test_packed_batch_matches_torch_packed_sequence():
    """PackedBatch (the index bookkeeping in front of the recurrent kernels) reproduces
    pack_padded_sequence(enforce_sorted=False) / pad_packed_sequence, the calls of
    rnn_dyn/RNNWrapper.py:89-102: same row order, same per-step batch sizes, same padding; and its
    shifted-frame index gives h_{t-1} of every packed frame.  The tables are applied here with the
    definition of the native row gather (out[r] = src[idx[r]], the fill row where idx[r] is negative
    or one past the end); the kernel itself: tests/test_gpu_nn.py::test_rows_gather_is_pack_unpack_and_shift."""
    import torch
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from idiaptts_amd.nn.functional import PackedBatch

    def gather(src, idx, fill=None):
        fill = torch.zeros(src.shape[1]) if fill is None else fill
        ext = torch.cat([src, fill[None, :]])
        return ext[torch.where((idx < 0) | (idx >= src.shape[0]), torch.tensor(src.shape[0]), idx)]
    g = torch.Generator().manual_seed(5)
    for batch_first in (False, True):
        for lens in ([5, 9, 1, 9, 3], [4], [2, 2, 2], list(range(1, 18))):
            B, T = len(lens), max(lens) + 2                   # padded beyond the longest row
            x = torch.randn((B, T, 3) if batch_first else (T, B, 3), generator=g)
            pb = PackedBatch(lens, T, batch_first, "cpu")
            ref = pack_padded_sequence(x, torch.tensor(lens), batch_first=batch_first,
                                       enforce_sorted=False)
            packed = gather(x.reshape(-1, 3), pb.flat_index)
            assert torch.equal(packed, ref.data)
            sizes = np.diff(np.concatenate([pb.d_row_off.numpy(), [pb.N]]))
            assert list(sizes) == list(ref.batch_sizes.numpy())
            assert torch.equal(pb.perm, ref.sorted_indices) and pb.T == max(lens)
            back, _ = pad_packed_sequence(ref, batch_first=batch_first, total_length=T)
            assert torch.equal(gather(packed, pb.inv_flat).reshape(x.shape), back)
            # reverse-direction row table and the h_{t-1} shift, against explicit loops
            sl = pb.h_lengths.numpy()
            off = pb.d_row_off.numpy()
            rev = pb.d_rev_row.numpy()
            y = torch.arange(pb.N * 2, dtype=torch.float32).reshape(pb.N, 2) + 1.0
            h0 = torch.tensor([[-1.0], [-2.0]])
            hp = torch.cat([gather(y[:, d:d + 1], pb.prev_row[d], h0[d]) for d in range(2)], dim=1)
            for b in range(B):
                for t in range(sl[b]):
                    s = sl[b] - 1 - t
                    assert rev[s, b] == off[t] + b
                    r = off[t] + b
                    want_f = y[off[t - 1] + b, 0] if t > 0 else h0[0, 0]
                    want_r = y[off[t + 1] + b, 1] if t + 1 < sl[b] else h0[1, 0]
                    assert hp[r, 0] == want_f and hp[r, 1] == want_r
    with pytest.raises(ValueError):
        PackedBatch([3, 0], 3, False, "cpu")


def test_frame_shard_round_trip_and_gather(tmp_path):
    from idiaptts_amd.src.data_preparation.FrameShard import FrameShard
    rng = np.random.default_rng(0)
    lens = (3, 7, 2, 5)
    xs = [rng.normal(size=(t, 5)).astype(np.float32) for t in lens]
    ys = [rng.normal(size=(t, 3)).astype(np.float32) for t in lens]
    shard = FrameShard.from_arrays(xs, ys, ["a", "b", "c", "d"], {"norm": "min_max"})
    assert shard.x.shape == (17, 8) and shard.y.shape == (17, 4)       # pitches padded to 4 floats
    assert np.all(shard.x[:, 5:] == 0) and np.all(shard.y[:, 3:] == 0)
    path = shard.save(str(tmp_path / "train.ittshard"))
    for mmap in (True, False):
        back = FrameShard.load(path, mmap=mmap)
        assert back.ids == ["a", "b", "c", "d"] and back.meta == {"norm": "min_max"}
        assert list(back.offsets) == [0, 3, 10, 12, 17] and list(back.lengths) == list(lens)
        x, y, l = back.gather([3, 0, 1])
        assert list(l) == [5, 3, 7]
        assert np.array_equal(x[:, :5], np.concatenate([xs[3], xs[0], xs[1]]))
        assert np.array_equal(y, np.concatenate([ys[3], ys[0], ys[1]]))
    dev = shard.to("cpu")                      # torch-resident variant (the device path)
    x, y, _ = dev.gather([2])
    assert np.array_equal(x[:, :5].numpy(), xs[2]) and y.shape == (2, 3) and y.stride(0) == 4
    with open(path, "r+b") as f:
        f.write(b"XXXX")
    with pytest.raises(ValueError):
        FrameShard.load(path)


def test_question_labels_from_hts_labels_are_bit_exact(golden_dir, tmp_path):
    """HTS full-context labels + question file -> the frame-level question labels
    the reference's fixtures hold (test/integration/fixtures/questions/*.questions, generated by
    its label_normalisation.py), bit for bit, and the min / max over them."""
    import zipfile
    from idiaptts_amd.src.data_preparation.questions.QuestionLabelGen import QuestionLabelGen
    from idiaptts_amd.src.data_preparation.questions.label_normalisation import \
        HTSLabelNormalisation, wildcards_to_regex
    lab_dir = str(tmp_path / "lab")
    zipfile.ZipFile(os.path.join(golden_dir, "labels_state_align.zip")).extractall(lab_dir)
    qfile = os.path.join(golden_dir, "questions-en-radio_dnn_400.hed")
    g = np.load(os.path.join(golden_dir, "trainer_fixture.npz"))
    ids = [str(i) for i in g["id_list"]]
    out_dir = str(tmp_path / "questions")
    labels, qmin, qmax = QuestionLabelGen.gen_data(lab_dir, qfile, out_dir, "file_id_list.txt",
                                                   ids, return_dict=True)
    assert list(labels) == ids
    for i in ids:
        saved = np.load(os.path.join(out_dir, i + ".npz"))["questions"]
        assert saved.dtype == np.float32 and np.array_equal(saved, g["questions/" + i]), i
        assert np.array_equal(labels[i].astype(np.float32), saved)
    mm = np.load(os.path.join(out_dir, "file_id_list-min-max.npz"))
    ref = np.frombuffer(g["bin/questions/min-max.bin"].tobytes(), dtype=np.float64).reshape(2, -1)
    assert np.array_equal(mm["min"], qmin) and np.array_equal(mm["max"], qmax)
    assert np.array_equal(qmin, ref[0]) and np.array_equal(qmax, ref[1])
    # the reader side reads what gen_data wrote
    reader = QuestionLabelGen(out_dir, 409)
    reader.get_normalisation_params(out_dir, "file_id_list")
    x = reader[ids[1]]["questions"]
    assert x.dtype == np.float32 and x.min() >= 0.0 and x.max() <= 1.0
    # pattern translation: anchoring and the numeric capture group
    assert wildcards_to_regex("*-aa+*") == r"\-aa\+"
    assert wildcards_to_regex("aa~*") == r"\Aaa\~" and wildcards_to_regex("*|1") == r"\|1\Z"
    assert wildcards_to_regex(r"@(\d+)_", True) == r"@(\d+)_"
    h = HTSLabelNormalisation(qfile)
    assert (h.dict_size, h.dimension) == (400, 409)
    with pytest.raises(NotImplementedError):
        HTSLabelNormalisation(qfile, subphone_feats="none")


def test_phoneme_and_duration_readers_match_reference(golden_dir, tmp_path):
    """Config 4 inputs: PhonemeLabelGen ids for both label layouts equal what the reference's
    reader returned (tests/golden/make_golden.py --duration); the duration reader reads the
    legacy fixtures, normalises with the legacy mean-std_dev.bin and regenerates the same
    durations from the state-aligned labels."""
    from fixture_dirs import materialise_duration
    from idiaptts_amd.src.Metrics import Metrics
    from idiaptts_amd.src.data_preparation.phonemes.PhonemeDurationLabelGen import \
        PhonemeDurationLabelGen
    from idiaptts_amd.src.data_preparation.phonemes.PhonemeLabelGen import PhonemeLabelGen
    root = str(tmp_path)
    ids, g = materialise_duration(golden_dir, root)
    plist = os.path.join(root, "labels", "mono_phone.list")
    sd = PhonemeLabelGen.get_symbol_dict(plist)
    assert list(sd.keys()) == [str(s) for s in g["symbols"]] and sd["EOF"] == len(sd) - 1
    for ltype, sub in (("full_state_align", "label_state_align"), ("mono_no_align", "mono_no_align")):
        reader = PhonemeLabelGen(os.path.join(root, "labels", sub), plist, label_type=ltype,
                                 one_hot=True)
        for i in ids:
            raw = reader.load(i)
            assert raw.dtype == np.int64 and np.array_equal(raw, g["ids_%s/%s" % (ltype, i)])
            oh = reader[i]["phonemes"]
            assert oh.shape == (len(raw), len(sd)) and np.array_equal(oh.argmax(1), raw[:, 0])
    with pytest.raises(AssertionError):
        PhonemeLabelGen(os.path.join(root, "labels", "label_state_align"), plist,
                        label_type="HTK full").load(ids[0])
    eof = PhonemeLabelGen(os.path.join(root, "labels", "mono_no_align"), plist,
                          label_type="mono_no_align", add_EOF=True)
    x = eof[ids[0]]["phonemes"]
    assert x[-1, 0] == sd["EOF"] and np.array_equal(eof.postprocess_sample(x), x[:-1])
    # durations: legacy files, legacy normalisation parameters, regeneration from labels
    dreader = PhonemeDurationLabelGen(os.path.join(root, "dur"))
    mean, std = dreader.get_normalisation_params()
    assert mean.shape[-1] == 5 and std.shape[-1] == 5
    for i in ids:
        d = dreader.load(i)
        assert np.array_equal(d, g["dur/" + i])
        gen = PhonemeDurationLabelGen._get_full_state_align_dur(
            os.path.join(root, "labels", "label_state_align", i + ".lab"))
        assert np.array_equal(gen, d)
        n = dreader[i]["durations"]
        assert n.dtype == np.float32 and np.allclose(dreader.postprocess_sample(n), d, atol=1e-5)
    assert np.array_equal(Metrics.rmse(g["metric_a"], g["metric_b"]), g["metric_rmse"])
    assert np.abs(Metrics.pearson(g["metric_a"], g["metric_b"]) - g["metric_pearson"]).max() < 1e-12


def test_npz_data_reader_matches_reference_fixture(golden_dir, tmp_path):
    """NpzDataReader (reference data_preparation/NpzDataReader.py:140-420): archives in one or
    several directories, index subsets, the three normalisers from files or given, pre/post
    functions on either side of the normalisation, chunk padding -- outputs captured from the
    reference's reader on the same seeded archives (tests/golden/make_golden.py --npz-reader)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden",
                                                  os.path.join(golden_dir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from idiaptts_amd.src.data_preparation.NpzDataReader import DataReader, NpzDataReader
    got = mg.run_npz_reader_cases(NpzDataReader, str(tmp_path))
    want = np.load(os.path.join(golden_dir, "npz_reader_fixture.npz"))
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        assert np.array_equal(got[k], want[k]), k
    # error behaviour of the reference
    reader = NpzDataReader.Config(name="cmp", directory=str(tmp_path / "a")).create_reader()
    with pytest.raises(FileNotFoundError):
        reader.load("nope")
    with pytest.raises(RuntimeError):
        NpzDataReader.Config(name="x", directory=str(tmp_path / "a"), features=["cmp", "dur"],
                             output_names=["only_one"]).create_reader()["utt1"]
    normed = NpzDataReader.Config(name="cmp", directory=str(tmp_path / "a"),
                                  norm_type=NpzDataReader.Config.NormType.MEAN_STDDEV).create_reader()
    with pytest.raises(ValueError):
        normed["utt1"]                                  # parameters not loaded yet
    with pytest.raises(NotImplementedError):
        DataReader(DataReader.Config("base")).load("utt1")


def test_native_question_labels_equal_the_regex_formulation(golden_dir, tmp_path):
    """csrc/labels.cpp (token lists + backtracking matcher, the production path) against the
    numpy / `re` formulation it replaces: identical label matrices on the fixture `.lab` files, and
    identical answers on adversarial context strings / patterns (anchors, inner wildcards, the two
    CQS capture groups, backtracking into the digit group, 'LL-' questions)."""
    import ctypes
    import zipfile
    from idiaptts_amd import lib as _lib
    from idiaptts_amd.src.data_preparation.questions.label_normalisation import \
        HTSLabelNormalisation, QuestionSet
    lab_dir = str(tmp_path / "lab")
    zipfile.ZipFile(os.path.join(golden_dir, "labels_state_align.zip")).extractall(lab_dir)
    qfile = os.path.join(golden_dir, "questions-en-radio_dnn_400.hed")
    g = np.load(os.path.join(golden_dir, "trainer_fixture.npz"))
    ids = [str(i) for i in g["id_list"]]
    nat = HTSLabelNormalisation(qfile, n_threads=3)
    ref = HTSLabelNormalisation(qfile, native=False)
    assert nat._handle is not None and ref._handle is None
    files = [os.path.join(lab_dir, i + ".lab") for i in ids]
    block, off = nat.generate_batch(files)
    for k, f in enumerate(files):
        want = ref.load_labels_with_state_alignment(f)
        assert np.array_equal(block[off[k]:off[k + 1]], want), f
        assert np.array_equal(nat.load_labels_with_state_alignment(f), want)
    # perform_normalisation works in chunks of files (bounded memory): any chunk size gives the same
    # archives, dictionary and min / max as the file-by-file formulation
    import os as _os
    want_dict, want_params = ref.perform_normalisation("ids.txt", ids, lab_dir, str(tmp_path / "q_ref"),
                                                       return_dict=True)
    for chunk in ("2", "256"):
        _os.environ["ITTS_LABEL_CHUNK_FILES"] = chunk
        try:
            got_dict, got_params = nat.perform_normalisation("ids.txt", ids, lab_dir,
                                                             str(tmp_path / ("q_nat" + chunk)), return_dict=True)
        finally:
            del _os.environ["ITTS_LABEL_CHUNK_FILES"]
        assert list(got_dict) == list(want_dict)
        for k in want_dict:
            assert np.array_equal(got_dict[k], want_dict[k].astype(np.float32))
            assert np.array_equal(np.load(str(tmp_path / ("q_nat" + chunk) / (k + ".npz")))["questions"],
                                  np.load(str(tmp_path / "q_ref" / (k + ".npz")))["questions"])
        for a, b in zip(got_params, want_params):
            assert np.array_equal(np.asarray(a), np.asarray(b))
    # adversarial question file
    qs = tmp_path / "q.hed"
    qs.write_text("\n".join([
        'QS "C-a" {*-a+*,*-aa+*}', 'QS "LL-x" {x^*,yy^*}', 'QS "exact" {abc}', 'QS "pre" {ab*}',
        'QS "suf" {*yz}', 'QS "mid" {a*c*e}', 'QS "all" {*}', 'QS "dots" {*a.b*}',
        'QS "paren" {*(x)*}', 'QS "two" {p*q,*r|s*}', 'QS "LL-lit" {lit}',
        'CQS "num" {@(\\d+)_}', 'CQS "numdot" {*:([\\d\\.]+)+*}', 'CQS "back" {#(\\d+)5}',
        'CQS "anch" {(\\d+)=*}', 'CQS "end" {*/E:(\\d+)}', '']))
    L = _lib.load()
    handle, nb, nc = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
    _lib.check(L.itts_questions_load(os.fsencode(str(qs)), ctypes.byref(handle), ctypes.byref(nb),
                                     ctypes.byref(nc)), "load")
    py = QuestionSet(str(qs))
    assert (nb.value, nc.value) == (len(py.binary), len(py.continuous)) == (11, 5)
    rng = np.random.default_rng(0)
    alphabet = list("abcxyz^-+|@_:#=/E.()0123456789prqslit")
    labels = ["abc", "ab", "xabcx", "x^a-aa+b", "yy^", "ayy^", "aXcXe", "ace", "a.b", "aXb",
              "(x)", "p..q", "r|s", "lit", "alit", "@12_", "@_", "x@007_y", ":1.5+", ":..+",
              "#1255", "#55", "#5", "12=rest", "a12=", "/E:3", "/E:3x", "", "a-a+"]
    labels += ["".join(rng.choice(alphabet, size=int(rng.integers(0, 14)))) for _ in range(3000)]
    out = np.empty(16)
    for lab in labels:
        try:
            want = py.vector(lab)
        except ValueError:          # float('1..2'): the reference raises there as well
            continue
        _lib.check(L.itts_questions_vector(handle, lab.encode(), out.ctypes.data), "vector")
        assert np.array_equal(out, want), (lab, out, want)
    L.itts_questions_free(handle)
    # malformed inputs come back as errors with a message
    bad = tmp_path / "bad.lab"
    bad.write_text("0 50000 a-b+c[2]\n50000 100000 a-b+c[3]\n")
    with pytest.raises(_lib.IttsError):
        nat.generate_batch([str(bad)])
    with pytest.raises(_lib.IttsError):
        nat.generate_batch([str(tmp_path / "missing.lab")])
