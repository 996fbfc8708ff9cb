"""Hyper-parameter container with the reference's attribute names and defaults
(idiaptts/src/ExtendedHParams.py:20-131 for the container API, :133-330 for the defaults).
The reference derives from a tensorflow-style HParams; this one is a plain attribute bag with the
same helper surface (add_hparams / add_hparam / set_hparam / setattr_no_type_check / has_value /
get_value / verify / get_debug_string / parse / values)."""
import logging


class ExtendedHParams(object):

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            object.__setattr__(self, k, v)

    # ------------------------------------------------------------------ container API
    def add_hparam(self, name, value):
        if name in self.__dict__:
            raise ValueError("Hyperparameter name is reserved: %s" % name)
        object.__setattr__(self, name, value)

    def add_hparams(self, **kwargs):
        """Adds new or overrides existing values (reference :67-72)."""
        for k, v in kwargs.items():
            object.__setattr__(self, k, v)

    def set_hparam(self, name, value):
        object.__setattr__(self, name, value)

    def setattr_no_type_check(self, name, value):
        object.__setattr__(self, name, value)

    def has_value(self, attribute):
        return getattr(self, attribute, None) is not None

    def get_value(self, attribute, default=None):
        return getattr(self, attribute) if self.has_value(attribute) else default

    def values(self):
        return dict(self.__dict__)

    def override_from_hparam(self, hparam):
        for k, v in hparam.values().items():
            object.__setattr__(self, k, v)
        return self

    def verify(self):
        """The reference warns about attributes that were set without add_hparam; a plain
        attribute bag has nothing to verify."""
        return True

    def enable_backwards_compatibility(self):
        """Maps deprecated names onto the current ones (reference :108-130)."""
        if self.get_value("load_from_checkpoint", False) and self.has_value("checkpoint_epoch"):
            self.load_checkpoint_epoch = self.checkpoint_epoch
        if self.has_value("checkpoint_step") and not self.has_value("load_checkpoint_step"):
            self.load_checkpoint_step = self.checkpoint_step
        if self.has_value("learning_rate") and "lr" not in self.optimiser_args:
            self.optimiser_args["lr"] = self.learning_rate

    def get_debug_string(self, max_array_elements=15):
        lines = []
        for k in sorted(self.__dict__):
            v = self.__dict__[k]
            if hasattr(v, "__len__") and not isinstance(v, (str, dict)) and \
                    len(v) > max_array_elements:
                v = "{}... ({} elements)".format(list(v[:max_array_elements]), len(v))
            lines.append("  {}={}".format(k, v))
        return "Hyper-parameters:\n" + "\n".join(lines)

    def parse(self, hparams_string):
        """`name=value,name=value` with literal evaluation of the values."""
        import ast
        for item in filter(None, (s.strip() for s in hparams_string.split(","))):
            name, value = item.split("=", 1)
            try:
                value = ast.literal_eval(value)
            except (ValueError, SyntaxError):
                pass
            object.__setattr__(self, name.strip(), value)
        return self

    # --------------------------------------------------------------------- defaults
    @staticmethod
    def create_hparams(hparams_string=None, verbose=False):
        hp = ExtendedHParams(
            # general
            voice=None, work_dir=None, data_dir=None, logging_batch_index_perc=10,
            start_with_test=True, log_memory_consumption=True, epochs_per_test=1,
            networks_dir="nn", checkpoints_dir="checkpoints", epochs_per_checkpoint=1,
            steps_per_checkpoint=0, save_final_model=True, use_best_as_final_model=True,
            gen_figure_ext=".pdf",
            # experiment
            epochs=0, test_set_perc=0.05, val_set_perc=0.05, seed=None, use_gpu=False,
            num_gpus=1, batch_first=False, shuffle_train_set=True, shuffle_val_set=False,
            batch_size_train=1, batch_size_test=48, batch_size_val=48, batch_size_benchmark=48,
            batch_size_synth=48, batch_size_gen_figure=48,
            dataset_type="PyTorchDatareadersDataset", dataset_num_workers_gpu=4, dataset_worker_kind="thread",
            dataset_num_workers_cpu=0, dataset_pin_memory=True, dataset_load_async=True,
            teacher_forcing_in_test=False, preload_next_batch_to_gpu=False,
            # not in the reference: keep the (normalised, length matched) training data resident in
            # HBM as packed frame shards instead of loading one file per item and step
            resident_dataset=False,
            # not in the reference: the rows the data readers return for an utterance stay in HBM after their first
            # use and every later mini-batch is gathered on the device (the same batches in the same order: data_
            # preparation/DeviceBatchCache.py); `dataset_device_cache_bytes` bounds what is kept (None: 60 % of the
            # memory free at the first upload), `dataset_host_cache_bytes` what page-locked host memory holds beyond that
            # (None: a quarter of the free host memory; these utterances cross PCIe per batch but are not read again)
            dataset_device_cache=True, dataset_device_cache_bytes=None, dataset_host_cache_bytes=None,
            # not in the reference: checkpoints are serialised by a background thread
            async_checkpoint=False,
            # data
            input_norm_params_file_prefix=None, output_norm_params_file_prefix=None,
            len_in_out_multiplier=1, out_dir=None, world_dir=None,
            # audio
            frame_size_ms=5,
            # model
            model_type=None, model_name=None, model_path=None, load_from_checkpoint=False,
            load_newest_checkpoint=False, load_checkpoint_epoch=None, load_checkpoint_step=None,
            checkpoint_epoch=None, checkpoint_step=None, load_optimiser=True,
            load_scheduler=True, ignore_layers=list(), layer_map=list(),
            allow_missing_layers=False, dropout=0.0, hidden_init=0.0, train_hidden_init=False,
            # optimisation
            loss_per_sample=False, backward_retain_graph=False, optimiser_type="Adam",
            optimiser_args=dict(), use_saved_learning_rate=True, replace_inf_grads_by_zero=False,
            ema_decay=None, scheduler_type="default", scheduler_args=dict(),
            iterations_per_scheduler_step=None, epochs_per_scheduler_step=None,
            grad_clip_norm_type=None, grad_clip_max_norm=None, grad_clip_thresh=None,
            optimiser=None, backprop_loss_names=None, scheduler=None, scheduler_loss_names=None,
            # synthesis
            synth_vocoder="WORLD", synth_vocoder_path=None, synth_ext="wav", synth_fs=16000,
            sp_type="mcep", num_coded_sps=60, num_bap=1, synth_dir=None,
            synth_acoustic_model_path=None, synth_file_suffix='', do_post_filtering=False,
            synth_gen_figure=False)
        if hparams_string:
            logging.info('Parsing command line hparams: %s', hparams_string)
            hp.parse(hparams_string)
        if verbose:
            logging.info(hp.get_debug_string())
        return hp
