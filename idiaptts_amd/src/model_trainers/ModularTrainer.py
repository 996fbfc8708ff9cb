"""ModularTrainer: data split, model / loss set-up, the epoch loop with validation, scheduler and
checkpoint policy, batched forwarding with post-processing, benchmark and waveform generation --
the reference's idiaptts/src/model_trainers/ModularTrainer.py with the same method names,
arguments and return values:

  __init__ / _setup_id_lists (:43-117)   init (:187-251)   train (:379-517)   test (:607-615)
  forward / synth / benchmark / gen_output (:617-792)   _forward_batched (:814-887)
  split_batch / _split_return_values (:127-185)   gen_waveform (:1014-1085)   copy_synth (:1093-1119)
  save_checkpoint / load_checkpoint / load_best_model / get_model_path (:305-368)

Everything that computes runs on the HIP kernels through the model handler, the data readers and
the Synthesiser; loss scalars go to TensorBoard when it is installed (idiaptts_amd/misc/logging_sinks.py:
a JSON-lines file otherwise), figure generation is not part of the accelerated path."""
import copy
import logging
import os
import random
import resource
from datetime import datetime, timedelta
from functools import partial
from timeit import default_timer as timer
from typing import Dict, List

import numpy as np

from idiaptts_amd.nn.functional import padding_rows_identical
from idiaptts_amd.misc import logging_sinks
from idiaptts_amd.src.data_preparation.PyTorchDatareadersDataset import \
    PyTorchDatareadersDataset
from idiaptts_amd.src.ExtendedHParams import ExtendedHParams
from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
    ModularModelHandlerPyTorch as ModelHandler
from idiaptts_amd.src.Synthesiser import Synthesiser


class ModularTrainer(object):
    logger = logging.getLogger(__name__)

    def __init__(self, hparams: ExtendedHParams, id_list: List[str], data_reader_configs=None):
        assert hparams is not None
        self.tb_writer = None
        hparams.enable_backwards_compatibility()
        self._check_gpus(hparams)
        if hparams.seed is not None:
            ModelHandler.seed(hparams.seed)
            np.random.seed(hparams.seed)
            random.seed(hparams.seed)
        self._setup_id_lists(id_list, hparams)
        self.model_handler = ModelHandler()
        self.batch_collate_fn = None
        self.batch_decollate_fn = self.split_batch
        self.train_losses = []
        self.validation_losses = []
        self.reset_best_loss()
        self.total_epoch = 0
        self.total_steps = 0
        self.loss_modules = None
        self._data_reader_configs = data_reader_configs   # legacy: configs given up front

    def _check_gpus(self, hparams):
        """One process drives one GPU here; num_gpus > 1 means one process per GPU launched by
        torch.distributed.run (idiaptts_amd.parallel), so the device count is not asserted
        against hparams.num_gpus as the reference's DataParallel set-up does (:80-93)."""
        if hparams.use_gpu and not ModelHandler.cuda_is_available():
            raise RuntimeError("hparams.use_gpu is set but no GPU is visible; the HIP path has "
                               "no CPU fallback.")

    def _setup_id_lists(self, id_list, hparams):
        """[val | train | test] partition of the (seeded-shuffled) ids (reference :95-117)."""
        if getattr(self, "id_list_train", None) is not None:
            return
        shuffled = id_list
        if hparams.seed is not None:
            shuffled = random.sample(id_list, len(id_list))
        assert hparams.test_set_perc + hparams.val_set_perc < 1
        num_val = num_test = 0
        self.id_list_val = self.id_list_test = None
        if hparams.val_set_perc > 0.0:
            num_val = max(1, int(len(shuffled) * hparams.val_set_perc))
            self.id_list_val = shuffled[:num_val]
        if hparams.test_set_perc > 0.0:
            num_test = max(1, int(len(shuffled) * hparams.test_set_perc))
            self.id_list_test = shuffled[-num_test:]
        self.id_list_train = shuffled[num_val:-num_test] if num_test > 0 else shuffled[num_val:]
        assert len(self.id_list_train) > 0

    def reset_best_loss(self):
        self.best_loss = np.nan

    @staticmethod
    def create_hparams(hparams_string=None, verbose=False):
        return ExtendedHParams.create_hparams(hparams_string, verbose)

    # ------------------------------------------------------------------- batch decollation
    @staticmethod
    def split_batch(data: Dict[str, np.ndarray], seq_lengths: Dict[str, int], batch_first=True):
        return {k: ModularTrainer._split_return_values(v, seq_lengths.get(k),
                                                       batch_first=batch_first)
                for k, v in data.items()}

    @classmethod
    def _split_return_values(cls, input_values, seq_length_output, permutation=None,
                             batch_first=False):
        """Padded batch -> list of per-sample arrays cut to their lengths; tuples (hidden
        states) are split element-wise and regrouped per sample (reference :133-185)."""
        if input_values is None:
            return None
        if isinstance(input_values, tuple):
            if all(v is None for v in input_values):
                return input_values
            parts = tuple(cls._split_return_values(x, seq_length_output, permutation, batch_first)
                          for x in input_values)
            batch_size = len([t for t in parts if t is not None][0])
            return tuple(tuple(e if e is None or (isinstance(e, tuple)
                                                  and all(v is None for v in e)) else e[i]
                               for e in parts) for i in range(batch_size))
        if isinstance(input_values, list):
            return input_values
        if not isinstance(input_values, np.ndarray):
            cls.logger.error("Expected numpy tensor but input is of type {}."
                             .format(type(input_values)))
            raise TypeError()
        axis = 0 if batch_first else 1
        values = [np.squeeze(v, axis=axis)
                  for v in np.split(input_values, input_values.shape[axis], axis=axis)]
        if seq_length_output is not None and len(seq_length_output) > 1:
            values = [v[:seq_length_output[i]] for i, v in enumerate(values)]
        if permutation is not None:
            unsorted = list(values)
            for org_index, current_index in enumerate(permutation):
                unsorted[current_index] = values[org_index]
            values = unsorted
        return values

    def log_memory(self, use_gpu):
        """reference :253-256, with torch.cuda.mem_get_info in place of the nvidia-smi probe."""
        self.logger.info("CPU memory: {} MB.".format(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3))
        if use_gpu:
            self.logger.info("GPU memory: {} MB.".format(logging_sinks.get_gpu_memory_map()))

    # ---------------------------------------------------------------------------------- init
    def init(self, hparams, model_config=None, loss_configs=None, data_reader_configs=None):
        assert hparams.has_value("model_name"), "hparams.model_name is required."
        os.makedirs(os.path.join(hparams.out_dir, hparams.model_name, hparams.networks_dir),
                    exist_ok=True)
        self.log_memory(hparams.use_gpu)                                # reference :193
        self.tb_writer = logging_sinks.open_scalar_writer(hparams)      # reference :198-214
        self.datareaders = dict()
        if data_reader_configs is not None:
            self._data_reader_configs = copy.deepcopy(data_reader_configs)
        elif self._data_reader_configs is not None:
            self._data_reader_configs = copy.deepcopy(self._data_reader_configs)
        else:
            raise ValueError("Parameter data_reader_configs is required.")
        self._setup_datareaders(self._data_reader_configs, hparams)

        if hparams.load_newest_checkpoint:
            try:
                self.load_checkpoint(hparams)
            except FileNotFoundError:
                self.logger.warning("No newest checkpoint found, creating a new model instead.")
                self.create_model(model_config, use_gpu=hparams.use_gpu)
                if hparams.epochs > 0:
                    self.save_checkpoint(hparams)
        elif hparams.has_value("load_checkpoint_epoch") \
                or hparams.has_value("load_checkpoint_step"):
            self.load_checkpoint(hparams)
        else:
            if model_config is not None:
                self.create_model(model_config, use_gpu=hparams.use_gpu)
            else:
                assert self.model_handler.model is not None, \
                    "Model config is required or model must already be created"
            if hparams.epochs > 0:
                self.save_checkpoint(hparams)
        self._setup_loss_modules(loss_configs, use_gpu=hparams.use_gpu)
        self.logger.info("ModularTrainer ready.")

    def _setup_datareaders(self, data_reader_configs, hparams):
        readers = list()
        for config in data_reader_configs:
            reader = config.create_reader()
            readers.append(reader)
            for out_name in reader.output_names:
                self.datareaders[out_name] = reader
        self._readers = readers
        self.dataset_train = self.get_dataset(self.id_list_train, readers, hparams,
                                              is_train_set=True)
        self.dataset_val = self.dataset_test = None
        if self.id_list_val is not None:
            overlap = [i for i in self.id_list_val if i in self.id_list_train]
            assert len(overlap) == 0, "Found same ids in train and validation set: " \
                + ", ".join(overlap)
            self.dataset_val = self.get_dataset(self.id_list_val, readers, hparams,
                                                is_val_set=True)
        if self.id_list_test is not None:
            overlap = [i for i in self.id_list_test if i in self.id_list_train]
            assert len(overlap) == 0, "Found same ids in train and test set: " \
                + ", ".join(overlap)
            self.dataset_test = self.get_dataset(self.id_list_test, readers, hparams,
                                                 is_test_set=True)

    def _unique_readers(self):
        return list(dict.fromkeys(self.datareaders.values()))

    def get_dataset(self, id_list, datareaders, hparams, is_train_set=False, is_val_set=False,
                    is_test_set=False):
        if hparams.dataset_type != "PyTorchDatareadersDataset":
            raise NotImplementedError("Dataset type {} is not implemented."
                                      .format(hparams.dataset_type))
        return PyTorchDatareadersDataset(id_list, datareaders, hparams,
                                         is_train_set=is_train_set, is_val_set=is_val_set,
                                         is_test_set=is_test_set)

    def create_model(self, model_config, use_gpu):
        assert model_config is not None, "Model config is required."
        self.model_handler.create_model(model_config, use_gpu=use_gpu)
        self.total_epoch = 0
        self.total_steps = 0

    # --------------------------------------------------------------------------- checkpoints
    def save_checkpoint(self, hparams, model_path=None, save_as_best_model=False,
                        save_as_last_model=False):
        if model_path is None:
            model_path = self.get_model_path(hparams, ignore_model_path=True)
        self.model_handler.save_checkpoint(
            model_path=model_path, best_loss=self.best_loss, epoch=self.total_epoch,
            step=self.total_steps, save_as_best_model=save_as_best_model,
            save_as_epoch=hparams.epochs_per_checkpoint > 0,
            save_as_last_model=save_as_last_model,
            save_as_step=hparams.steps_per_checkpoint > 0)

    @staticmethod
    def get_model_path(hparams, ignore_model_path=False):
        if hparams.model_path is None or ignore_model_path:
            assert hparams.out_dir is not None and hparams.networks_dir is not None
            assert hparams.model_name is not None, \
                "A model_name has to be given. No default exists."
            return os.path.join(hparams.out_dir, hparams.model_name, hparams.networks_dir)
        return hparams.model_path

    def load_best_model(self, hparams, model_path=None):
        if model_path is None:
            model_path = self.get_model_path(hparams, ignore_model_path=True)
        try:
            self.best_loss, self.total_epoch, self.total_steps = \
                self.model_handler.load_checkpoint(
                    hparams=hparams, model_path=model_path, ignore_layers=False,
                    load_optimiser=True, load_scheduler=True, load_best_model=True)
            self.model_handler.ema = None
            self.logger.info("Using best (epoch {}) as final model.".format(self.total_epoch))
        except FileNotFoundError:
            self.logger.warning("No best model exists. Continue with current one.")

    def load_checkpoint(self, hparams, model_path=None):
        if model_path is None:
            model_path = self.get_model_path(hparams)
        try:
            self.best_loss, self.total_epoch, self.total_steps = \
                self.model_handler.load_checkpoint(
                    hparams=hparams, model_path=model_path,
                    epoch=hparams.get_value("load_checkpoint_epoch"), ignore_layers=True,
                    load_optimiser=hparams.load_optimiser, load_scheduler=hparams.load_scheduler,
                    step=hparams.get_value("load_checkpoint_step"))
        except FileNotFoundError as e:
            self.logger.error("Model does not exist at {}. {}".format(model_path, e))
            raise

    def _setup_loss_modules(self, loss_configs, use_gpu=False):
        if loss_configs is None:
            return
        if type(loss_configs) not in [tuple, list]:
            loss_configs = [loss_configs]
        self.loss_modules = [config.create_loss() for config in loss_configs]
        if use_gpu:
            self.loss_modules = [loss.cuda() for loss in self.loss_modules]

    # ---------------------------------------------------------------------------------- train
    def train(self, hparams):
        """Returns (validation losses, training losses, model_handler); the loss containers map
        loss name -> list over epochs (reference :379-517)."""
        self.sanity_check_train(hparams)
        self.logger.info(hparams.get_debug_string())
        if hparams.epochs <= 0:
            self.logger.info("Number of training epochs is {}. Skipping training."
                             .format(hparams.epochs))
            return list(), list(), self.model_handler
        if self.tb_writer is not None:                                  # reference :391-392
            self.tb_writer.add_text("HParams", "<pre>" + hparams.get_debug_string() + "</pre>")
        self.logger.info("Training set size: {}".format(len(self.id_list_train)))
        self.log_memory(hparams.use_gpu)                                # reference :422

        handler = self.model_handler
        handler.async_checkpoint = bool(hparams.get_value("async_checkpoint", False))
        handler.set_dataset(hparams, self.dataset_train, self.dataset_val, self.batch_collate_fn)
        handler.set_optimiser(hparams)
        handler.set_scheduler(hparams,
                              self.total_epoch if hparams.use_saved_learning_rate else 0,
                              self.total_steps if hparams.use_saved_learning_rate else 0)
        handler.set_losses(self.loss_modules)

        start_epoch, start_step = self.total_epoch, self.total_steps
        steps_per_training_epoch = len(handler.dataloader_train) // hparams.batch_size_train
        t_start = timer()
        self.logger.info('Start training: {}'.format(datetime.now().strftime("%Y-%m-%d %H:%M:%S")))

        if hparams.start_with_test:
            loss_dict = handler.test(hparams=hparams, total_epoch=start_epoch,
                                     total_steps=start_step, current_epoch=start_epoch)
            scheduler_loss = handler.get_summed_losses_subset(
                losses=loss_dict, loss_names=hparams.scheduler_loss_names)
            if np.isnan(self.best_loss) or scheduler_loss < self.best_loss:
                self.best_loss = scheduler_loss
            self.record_validation_loss(loss_dict, self.total_epoch)

        for current_epoch in range(1, hparams.epochs + 1):
            self.logger.info('Train epoch [{}/{}], step [{}/{}]:'.format(
                self.total_epoch + 1, start_epoch + hparams.epochs, self.total_steps + 1,
                start_step + hparams.epochs * steps_per_training_epoch))
            loss_dict = handler.train(hparams=hparams, total_epoch=self.total_epoch,
                                      total_steps=self.total_steps, current_epoch=current_epoch)
            self.total_epoch += 1
            self.total_steps += steps_per_training_epoch
            if self._has_nan_loss(loss_dict):
                break
            self.record_train_loss(loss_dict, self.total_epoch)

            current_model_saved = False
            if self.total_epoch % hparams.epochs_per_test == 0:
                loss_dict = handler.test(hparams=hparams, total_epoch=self.total_epoch,
                                         total_steps=self.total_steps,
                                         current_epoch=current_epoch)
                if self._has_nan_loss(loss_dict):
                    break
                self.record_validation_loss(loss_dict, self.total_epoch)
                scheduler_loss = handler.get_summed_losses_subset(
                    losses=loss_dict, loss_names=hparams.scheduler_loss_names)
                scheduler_epoch = self.total_epoch if hparams.use_saved_learning_rate \
                    else current_epoch
                handler.run_scheduler(hparams=hparams, loss=scheduler_loss,
                                      current_epoch=scheduler_epoch)
                if hparams.out_dir is not None:
                    if np.isnan(self.best_loss) or scheduler_loss < self.best_loss:
                        self.best_loss = scheduler_loss
                        self.save_checkpoint(hparams=hparams, save_as_best_model=True)
                        current_model_saved = True
                    if hparams.epochs_per_checkpoint > 0 \
                            and self.total_epoch % hparams.epochs_per_checkpoint == 0:
                        self.save_checkpoint(hparams=hparams)
                        current_model_saved = True
            if hparams.out_dir is not None and hparams.load_newest_checkpoint \
                    and not current_model_saved:
                self.save_checkpoint(hparams=hparams, save_as_last_model=True)

        self.logger.info('Training time: ' + str(timedelta(seconds=timer() - t_start)))
        self.log_losses(start_epoch=start_epoch)
        if hparams.out_dir is not None:
            if hparams.use_best_as_final_model:
                self.load_best_model(hparams)
            if hparams.save_final_model:
                self.save_checkpoint(hparams)
        self.model_handler.wait_for_checkpoints()     # everything on disk when train() returns
        return (*self.get_losses(), self.model_handler)

    def sanity_check_train(self, hparams):
        assert self.model_handler is not None and self.model_handler.model is not None, \
            "The init function has not been called before training."
        hparams.verify()
        if hparams.epochs_per_scheduler_step:
            if hparams.epochs_per_test > hparams.epochs_per_scheduler_step:
                self.logger.warning("Model is validated only every {} epochs, but scheduler is "
                                    "supposed to run every {} epochs.".format(
                                        hparams.epochs_per_test,
                                        hparams.epochs_per_scheduler_step))

    @staticmethod
    def _has_nan_loss(loss_dict):
        return any(np.isnan(v).any() for v in loss_dict.values())

    def record_train_loss(self, loss_dict, epoch):
        self.train_losses.append((loss_dict, epoch))

    def record_validation_loss(self, loss_dict, epoch):
        self.validation_losses.append((loss_dict, epoch))

    def _get_loss_names(self):
        if len(self.train_losses) > 0:
            return list(self.train_losses[0][0].keys())
        if len(self.validation_losses) > 0:
            return list(self.validation_losses[0][0].keys())
        return None

    def log_losses(self, start_epoch=-1):
        losses = self.get_losses(start_epoch)
        if losses is None:
            return
        for name in losses[0]:
            self.logger.info('Loss {} validation progress: '.format(name)
                             + ', '.join('{:.4f}'.format(l) for l in losses[0][name]))
            self.logger.info('Loss {} train progress: '.format(name)
                             + ', '.join('{:.4f}'.format(l) for l in losses[1][name]))

    def get_losses(self, start_epoch=-1):
        """({name: validation losses}, {name: training losses}) as arrays over the recorded
        epochs >= start_epoch (reference :590-605)."""
        names = self._get_loss_names()
        if names is None:
            return None
        val = {n: np.array([d[n] for d, e in self.validation_losses if e >= start_epoch])
               for n in names}
        train = {n: np.array([d[n] for d, e in self.train_losses if e >= start_epoch])
                 for n in names}
        return val, train

    def test(self, hparams):
        self.model_handler.set_dataset(hparams, self.dataset_train, self.dataset_val,
                                       self.batch_collate_fn)
        self.model_handler.set_losses(self.loss_modules)
        return self.model_handler.test(hparams=hparams, total_epoch=self.total_epoch,
                                       total_steps=self.total_steps,
                                       current_epoch=self.total_epoch)

    # ----------------------------------------------------------- forward / synth / benchmark
    def forward(self, hparams, ids_input, post_processing_mapping=None, load_target=True):
        """`load_target=False` reads only the streams the model consumes -- for ids that have no
        stored targets yet (text-to-speech: TTSModel.run_DM_AM)."""
        id_list = self._input_to_str_list(ids_input)
        return self._forward_batched(batch_size=hparams.batch_size_val, hparams=hparams,
                                     id_list=id_list,
                                     post_processing_mapping=post_processing_mapping,
                                     load_target=load_target)

    def synth(self, hparams, ids_input, post_processing_mapping=None, plotter_configs=None,
              load_target=True):
        id_list = self._input_to_str_list(ids_input)
        self.logger.info("Start synthesising [{0}]".format(", ".join(str(i) for i in id_list)))
        t_start = timer()
        out = self._forward_batched(batch_size=hparams.batch_size_synth, hparams=hparams,
                                    id_list=id_list,
                                    post_processing_mapping=post_processing_mapping,
                                    gen_figure=hparams.synth_gen_figure, synth=True,
                                    load_target=load_target)
        self.logger.info('Synthesis time for {} sample(s): {}'.format(
            len(id_list), timedelta(seconds=timer() - t_start)))
        return out

    def benchmark(self, hparams, post_processing_mapping=None, ids_input=None):
        """Scores of compute_score on the given ids, else on the test set, else on the
        validation set (reference :712-756)."""
        assert callable(getattr(self, 'compute_score', None)), \
            "Function has to be implemented for this trainer."
        if ids_input is None:
            if self.id_list_test is not None and len(self.id_list_test) > 0:
                id_list = sorted(self.id_list_test)
            elif self.id_list_val is not None and len(self.id_list_val) > 0:
                id_list = sorted(self.id_list_val)
            else:
                raise ValueError("No id list can be selected for benchmark, because non was "
                                 "given as parameter and test and validation set are empty.")
        else:
            id_list = self._input_to_str_list(ids_input)
        self.logger.info("Start benchmark on ({}): [{}]".format(
            len(id_list), ", ".join(str(i) for i in id_list)))
        return self._forward_batched(batch_size=hparams.batch_size_benchmark, hparams=hparams,
                                     id_list=id_list,
                                     post_processing_mapping=post_processing_mapping,
                                     benchmark=True)

    def gen_output(self, hparams, ids_input, post_processing_mapping=None):
        raise NotImplementedError("gen_output needs an OutputGen with save_output; use "
                                  "forward(...) and the label generator's save_output.")

    def gen_figure(self, *args, **kwargs):
        raise NotImplementedError("Figure generation is outside the accelerated path.")

    @staticmethod
    def _input_to_str_list(input):
        if isinstance(input, str):
            try:
                with open(input) as f:
                    return [s.strip(' \t\n\r') for s in f.readlines()]
            except IOError:
                return [input]
        if isinstance(input, (list, tuple)):
            return list(map(str, input))
        raise ValueError("Unknown input {} of type {}.".format(input, type(input)))

    def _forward_batched(self, batch_size, hparams, id_list, post_processing_mapping,
                         plotter_configs=None, load_target=True, synth=False, benchmark=False,
                         gen_figure=False):
        """Forward `id_list` in batches, split the padded outputs per id and run the mapped data
        reader's postprocess_sample on them (reference :814-887).  Returns (outputs,
        post-processed outputs), or the scores when `benchmark`."""
        assert len(id_list) > 0, "Received empty id_list."
        if gen_figure:
            raise NotImplementedError("Figure generation is outside the accelerated path.")
        readers = self._unique_readers()
        if not load_target:
            wanted = set(self._model_input_names())
            readers = [r for r in readers if wanted & set(r.output_names)]
            assert readers, "No data reader provides the model inputs {}.".format(sorted(wanted))
        dataset = self.get_dataset(id_list=id_list, datareaders=readers, hparams=hparams)
        dataloader = self.model_handler._get_dataloader(
            batch_size=batch_size, dataset=dataset, batch_first=hparams.batch_first,
            common_divisor=1, collate_fn=self.batch_collate_fn, num_workers=0,
            pin_memory=hparams.dataset_pin_memory, shuffle=False)
        dict_outputs, dict_outputs_post = {}, {}
        # batches of the stock collate function pad with zeros (pad_sequence): the Linear groups may then run on the
        # valid rows and one representative padding row (nn/functional.py: padding_rows_identical)
        stock = self.batch_collate_fn is None and all(
            r.min_frames is None or getattr(r, "pad_mode", "constant") == "constant" for r in readers)
        for data, seq_lengths in dataloader:
            id_sub_list = data["_id_list"]
            with padding_rows_identical(stock):
                data, seq_lengths = self.model_handler.inference(data=data, hparams=hparams,
                                                                 seq_lengths=seq_lengths)
            outputs = self.batch_decollate_fn(data, seq_lengths, batch_first=hparams.batch_first)
            # post-processing per output stream; readers that can handle the whole mini-batch at
            # once (WorldFeatLabelGen: one MLPG launch per stream) get it in one call
            post = {}
            for feature_name, per_id in outputs.items():
                if post_processing_mapping is None or feature_name not in post_processing_mapping:
                    continue
                reader_name = post_processing_mapping[feature_name]
                if reader_name is None:
                    post[feature_name] = list(per_id)
                    continue
                reader = self._reader_by_name(reader_name)
                batch_fn = getattr(reader, "postprocess_sample_batch", None)
                if batch_fn is not None:
                    post[feature_name] = batch_fn(list(per_id))
                else:
                    post[feature_name] = [reader.postprocess_sample(f) for f in per_id]
            for idx, id_name in enumerate(id_sub_list):
                dict_outputs[id_name] = {k: v[idx] for k, v in outputs.items()}
                dict_outputs_post[id_name] = {k: v[idx] for k, v in post.items()}
        if benchmark:
            return self.compute_score(data=dict_outputs_post, output=dict_outputs,
                                      hparams=hparams)
        if synth:
            self.gen_waveform(data=dict_outputs_post, hparams=hparams, id_list=id_list)
        return dict_outputs, dict_outputs_post

    def _reader_by_name(self, name):
        for reader in self._unique_readers():
            if reader.name == name:
                return reader
        raise KeyError("Unknown data reader {}.".format(name))

    def _model_input_names(self):
        """Names of the streams the (wrapped) model reads, from its config."""
        config = self.model_handler.model_config
        if config is None:
            config = getattr(self.model_handler.model, "config", None)
        names = getattr(config, "input_names", None)
        assert names, "The model config does not name its inputs."
        return list(names)

    # ------------------------------------------------------------------------------ waveforms
    def gen_waveform(self, id_list, data, hparams, use_model_name=True, has_deltas=False):
        """reference :1014-1085; only the WORLD vocoder is on the accelerated path."""
        if type(next(iter(data.values()))) is dict:
            if hparams.has_value("synth_feature_names"):
                feature_names = hparams.synth_feature_names
                if type(feature_names) not in [list, tuple]:
                    feature_names = (feature_names,)
            else:
                feature_names = list(next(iter(data.values())).keys())
            data = {id_name: np.concatenate([features[n] for n in feature_names], axis=1)
                    for id_name, features in data.items()}
        if hparams.synth_vocoder != "WORLD":
            raise NotImplementedError("Unknown vocoder type {}.".format(hparams.synth_vocoder))
        return Synthesiser.run_world_synth(data, hparams, use_model_name=use_model_name,
                                           has_deltas=has_deltas, epoch=self.total_epoch,
                                           step=self.total_steps)

    def get_labels(self, reader_name, id_name):
        for reader in self._unique_readers():
            if reader.name == reader_name:
                return reader[id_name]
        raise KeyError(reader_name)

    def copy_synth(self, hparams, id_list):
        """Vocoder round trip of the stored (un-normalised) features, written with the suffix
        `_ref` (reference :1093-1119)."""
        assert hparams.has_value("synth_feature_names"), \
            "hparams.synth_feature_names has to be given."
        feature_names = hparams.synth_feature_names
        if type(feature_names) not in [list, tuple]:
            feature_names = (feature_names,)
        ids_input = self._input_to_str_list(id_list)
        hparams = copy.deepcopy(hparams)
        readers = [self.datareaders[name] for name in feature_names]
        data = {id_name: np.concatenate([reader.load(id_name) for reader in readers], axis=1)
                for id_name in ids_input}
        hparams.synth_file_suffix += "_ref"
        has_deltas = any(getattr(r, "add_deltas", False) for r in readers)
        return self.gen_waveform(ids_input, data, hparams, use_model_name=False,
                                 has_deltas=has_deltas)
