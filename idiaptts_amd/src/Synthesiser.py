"""Synthesiser.run_world_synth with the reference's interface (idiaptts/src/Synthesiser.py:38-106).

The reference walks the utterances one by one (decode_sp -> world_features_to_raw -> write);
here all utterances of the call are decoded (mgc2sp), aperiodicity-decoded and synthesised by
single batched GPU launches, then written. `synth_world_features` is the name BASELINE.json's
north-star uses for this entry point; it is provided as an alias.
"""
import logging
import os
from typing import Dict

import numpy as np
import scipy.io.wavfile

from .data_preparation.audio.AudioProcessing import AudioProcessing
from .data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen


def _has(hparams, name):
    if hasattr(hparams, "has_value"):
        return hparams.has_value(name)
    return getattr(hparams, name, None) is not None


class Synthesiser(object):
    SYNTH_SUB_DIR = "synth"

    @staticmethod
    def run_world_synth(synth_output: Dict[str, np.ndarray], hparams, epoch: int = None,
                        step: int = None, use_model_name: bool = True,
                        has_deltas: bool = False, return_waveforms: bool = False):
        """Run the WORLD synthesize method on every entry of synth_output (id -> features)."""
        fs = hparams.synth_fs
        fft_size = AudioProcessing.fs_to_frame_length(fs)
        save_dir = Synthesiser._get_synth_dir(hparams, use_model_name, epoch=epoch, step=step)
        ids, amp_sps, lf0s, vuvs, baps = [], [], [], [], []
        for id_name, output in synth_output.items():
            coded_sp, lf0, vuv, bap = WorldFeatLabelGen.convert_to_world_features(
                output, contains_deltas=has_deltas, num_coded_sps=hparams.num_coded_sps,
                num_bap=hparams.num_bap)
            amp_sp = AudioProcessing.decode_sp(
                coded_sp, hparams.sp_type, fs,
                post_filtering=getattr(hparams, "do_post_filtering", False)
            ).astype(np.double, copy=False)
            ids.append(id_name)
            amp_sps.append(amp_sp)
            lf0s.append(lf0)
            vuvs.append(vuv)
            baps.append(bap)
        args = dict()
        for attr in "preemphasis", "f0_silence_threshold", "lf0_zero":
            if hasattr(hparams, attr):
                args[attr] = getattr(hparams, attr)
        waveforms = WorldFeatLabelGen.world_features_to_raw_batch(amp_sps, lf0s, vuvs, baps, fs=fs,
                                                                  n_fft=fft_size, **args)
        for id_name, waveform in zip(ids, waveforms):
            logging.info("Synthesise {} with the WORLD vocoder.".format(id_name))
            file_name = (os.path.basename(id_name) + getattr(hparams, "synth_file_suffix", "")
                         + '_' + str(hparams.num_coded_sps) + hparams.sp_type + "_WORLD")
            file_path = os.path.join(save_dir, file_name)
            # soundfile.write(float) default subtype for .wav is PCM_16
            pcm = np.clip(np.rint(waveform * 32768.0), -32768, 32767).astype(np.int16)
            scipy.io.wavfile.write(file_path + ".wav", fs, pcm)
            if getattr(hparams, "synth_ext", "wav").lower() != 'wav':
                raise NotImplementedError("Only wav output is supported (pydub is out of scope).")
        if return_waveforms:
            return dict(zip(ids, waveforms))

    synth_world_features = run_world_synth

    @staticmethod
    def _get_synth_dir(hparams, use_model_name: bool = True, epoch: int = None, step: int = None):
        """reference :82-106"""
        if _has(hparams, "synth_dir"):
            save_dir = hparams.synth_dir
        else:
            parts = [hparams.out_dir] if _has(hparams, "out_dir") else [os.path.curdir]
            if use_model_name and _has(hparams, "model_name"):
                parts.append(hparams.model_name)
            parts.append(Synthesiser.SYNTH_SUB_DIR)
            if epoch is not None:
                parts.append("e" + str(epoch))
            elif step is not None:
                parts.append("s" + str(step))
            save_dir = os.path.join(*parts)
        os.makedirs(save_dir, exist_ok=True)
        logging.info("Selected {} as synthesis directory.".format(save_dir))
        return save_dir
