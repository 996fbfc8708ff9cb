"""AcousticModelTrainer: question labels in, WORLD features (with deltas) out, MLPG + objective
scores for benchmarking and the WORLD vocoder for synthesis -- the reference's
idiaptts/src/model_trainers/AcousticModelTrainer.py (legacy_support_init :64-129,
create_hparams :131-160, init :163-189, benchmark :303-314, synth :379-400, compute_score
:402-432, get_output_dict :434-455, synthesize :457-520, copy_synth :522-528)."""
import copy
import logging
import os
from typing import List

import numpy as np

from idiaptts_amd.src.data_preparation.DataReaderConfig import DataReaderConfig
from idiaptts_amd.src.data_preparation.DataReaders import chunk_padding
from idiaptts_amd.src.data_preparation.world.WorldFeatLabelGen import WorldFeatLabelGen
from idiaptts_amd.src.Metrics import Metrics
from idiaptts_amd.src.model_trainers.ModularTrainer import ModularTrainer
from idiaptts_amd.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
from idiaptts_amd.src.neural_networks.pytorch.models import rnn_dyn
from idiaptts_amd.src.neural_networks.pytorch.models.NamedForwardWrapper import \
    NamedForwardWrapper


class AcousticModelTrainer(ModularTrainer):
    logger = logging.getLogger(__name__)

    def __init__(self, hparams, id_list: List[str], data_reader_configs=None):
        if hparams is None:
            hparams = self.create_hparams()
            hparams.out_dir = os.path.curdir
        super().__init__(data_reader_configs=data_reader_configs, id_list=id_list,
                         hparams=hparams)
        if hparams.scheduler_type == "default":
            hparams.scheduler_type = "Plateau"
            hparams.add_hparams(plateau_verbose=True)

    @staticmethod
    def legacy_support_init(dir_world_features, dir_question_labels, id_list, num_questions,
                            hparams):
        """Arguments for __init__ from the two feature directories."""
        data_reader_configs = [
            DataReaderConfig(name="questions", feature_type="QuestionLabelGen",
                             directory=dir_question_labels, features="questions",
                             num_questions=num_questions, match_length=["cmp_features"]),
            WorldFeatLabelGen.Config(name="cmp_features", directory=dir_world_features,
                                     features=["cmp_mcep" + str(hparams.num_coded_sps)],
                                     output_names=["acoustic_features"],
                                     add_deltas=hparams.add_deltas,
                                     num_coded_sps=hparams.num_coded_sps,
                                     num_bap=hparams.num_bap, sp_type=hparams.sp_type,
                                     requires_seq_mask=True, match_length=["questions"])]
        hparams.world_dir = dir_world_features
        return dict(data_reader_configs=data_reader_configs, hparams=hparams, id_list=id_list)

    @staticmethod
    def create_hparams(hparams_string=None, verbose=False):
        hparams = ModularTrainer.create_hparams(hparams_string, verbose=False)
        hparams.add_hparams(
            num_questions=None, question_file=None, num_coded_sps=60, num_baps=1, load_sp=True,
            load_lf0=True, load_vuv=True, load_bap=True, sp_type="mcep", add_deltas=True,
            synth_load_org_sp=False, synth_load_org_lf0=False, synth_load_org_vuv=False,
            synth_load_org_bap=False,
            metrics=[Metrics.MCD, Metrics.F0_RMSE, Metrics.VDE, Metrics.BAP_distortion])
        if verbose:
            logging.info(hparams.get_debug_string())
        return hparams

    def init(self, hparams, data_reader_configs=None, model_config=None, loss_configs=None):
        if model_config is None and hparams.has_value("model_type"):
            model_config = NamedForwardWrapper.Config(
                wrapped_model_config=rnn_dyn.convert_legacy_to_config(
                    (hparams.num_questions,), hparams),
                input_names=["questions"], batch_first=hparams.batch_first,
                name="AcousticModel", output_names=["pred_acoustic_features"])
        if loss_configs is None:
            loss_configs = [NamedLoss.Config(
                name="MSELoss_acoustic_features", type_="MSELoss",
                seq_mask="acoustic_features_mask",
                input_names=["acoustic_features", "pred_acoustic_features"],
                batch_first=hparams.batch_first)]
        super().init(data_reader_configs=data_reader_configs, hparams=hparams,
                     model_config=model_config, loss_configs=loss_configs)
        self.logger.info("AcousticModelTrainer ready.")

    def benchmark(self, hparams, post_processing_mapping=None, ids_input=None):
        if post_processing_mapping is None:
            post_processing_mapping = {"pred_acoustic_features": "cmp_features"}
        return super().benchmark(hparams=hparams,
                                 post_processing_mapping=post_processing_mapping,
                                 ids_input=ids_input)

    def synth(self, hparams, ids_input, post_processing_mapping=None, plotter_configs=None,
              load_target=True):
        if post_processing_mapping is None:
            post_processing_mapping = {"pred_acoustic_features": "cmp_features",
                                       "acoustic_features": "cmp_features"}
        if not hparams.has_value("synth_feature_names"):
            hparams = copy.deepcopy(hparams)
            hparams.add_hparams(synth_feature_names=["pred_acoustic_features"])
        return super().synth(hparams=hparams, ids_input=ids_input,
                             post_processing_mapping=post_processing_mapping,
                             plotter_configs=plotter_configs, load_target=load_target)

    def compute_score(self, data, output, hparams):
        """{label name: (MCD, F0 RMSE, VDE, BAP distortion) averaged over the ids}; `data` holds
        the post-processed (de-normalised, MLPG-smoothed) network outputs per id, the originals
        are loaded from hparams.world_dir (reference :402-432)."""
        dict_original_post = self.get_output_dict(
            data.keys(), hparams, chunk_size=hparams.get_value("n_frames_per_step", default=1))
        metric_dict = {}
        for label_name in next(iter(data.values())).keys():
            per_id = []
            for id_name, labels in data.items():
                out = WorldFeatLabelGen.convert_to_world_features(
                    sample=labels[label_name], contains_deltas=False,
                    num_coded_sps=hparams.num_coded_sps, num_bap=hparams.num_bap)
                org = WorldFeatLabelGen.convert_to_world_features(
                    sample=dict_original_post[id_name], contains_deltas=hparams.add_deltas,
                    num_coded_sps=hparams.num_coded_sps, num_bap=hparams.num_bap)
                scores = Metrics.get_metrics(org, out)
                per_id.append(scores)
                self.logger.info("{} {}: MCD {:.3f} dB, F0 RMSE {:.2f} Hz, VDE {:.3f} %, "
                                 "BAP {:.3f} dB".format(label_name, id_name, *scores))
            metric_dict[label_name] = list(np.mean(np.array(per_id, dtype=np.float64), axis=0))
        return metric_dict

    def get_output_dict(self, id_list, hparams, chunk_size=1):
        assert hparams.has_value("world_dir"), \
            "hparams.world_dir must be set for this operation."
        out = dict()
        for id_name in id_list:
            sample = WorldFeatLabelGen.load_sample(
                id_name, dir_out=hparams.world_dir, add_deltas=hparams.add_deltas,
                num_coded_sps=hparams.num_coded_sps, sp_type=hparams.sp_type,
                num_bap=hparams.num_bap, load_sp=hparams.load_sp, load_lf0=hparams.load_lf0,
                load_vuv=hparams.load_vuv, load_bap=hparams.load_bap)
            if chunk_size > 1:
                sample = np.pad(sample, chunk_padding(sample, chunk_size), 'constant')
            out[id_name] = sample
        return out

    def synthesize(self, data, hparams, id_list):
        """Optionally overrides streams of the network output with the stored originals
        (hparams.synth_load_org_{sp,lf0,vuv,bap}), then runs the vocoder (reference :457-520;
        like there, synth() does not route through this method, callers use it directly)."""
        feature_names = hparams.get_value("synth_feature_names",
                                          list(next(iter(data.values())).keys()))
        if type(feature_names) not in [list, tuple]:
            feature_names = (feature_names,)
        for id_name, features in data.items():
            data[id_name] = np.concatenate([features[n] for n in feature_names], axis=1)
        if hparams.synth_load_org_sp or hparams.synth_load_org_lf0 \
                or hparams.synth_load_org_vuv or hparams.synth_load_org_bap:
            assert hparams.has_value("world_dir"), \
                "hparams.world_dir must be set for this operation."
            for id_name, labels in data.items():
                org = WorldFeatLabelGen.load_sample(id_name, hparams.world_dir,
                                                    num_coded_sps=hparams.num_coded_sps,
                                                    num_bap=hparams.num_bap)
                len_diff = len(org) - len(labels)
                if len_diff > 0:
                    org = WorldFeatLabelGen.trim_end_sample(org, int(len_diff / 2), reverse=True)
                    org = WorldFeatLabelGen.trim_end_sample(org, len_diff - int(len_diff / 2))
                n = len(org)
                if hparams.synth_load_org_sp:
                    labels[:n, :hparams.num_coded_sps] = org[:, :hparams.num_coded_sps]
                if hparams.synth_load_org_lf0:
                    labels[:n, -3] = org[:, -3]
                if hparams.synth_load_org_vuv:
                    labels[:n, -2] = org[:, -2]
                if hparams.synth_load_org_bap:
                    labels[:n, -hparams.num_bap:] = org[:, -hparams.num_bap:]
        return super().gen_waveform(id_list=id_list, data=data, hparams=hparams)

    def copy_synth(self, hparams, id_list):
        if not hparams.has_value("synth_feature_names"):
            hparams.setattr_no_type_check("synth_feature_names", "acoustic_features")
        return super().copy_synth(hparams=hparams, id_list=id_list)
